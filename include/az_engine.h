/* az_engine.h -- C ABI of the MI355X-native AlphaZero self-play engine.
 *
 * Drop-in boundary for the async_mcts + arena hot path of AnimatedRNG/alphazero-rs.
 * The reference has no FFI: its engine is reached through generic Rust traits
 * (`Game`, src/game.rs:10-28; `NNet`, src/nnet.rs:35-45) consumed by
 * `AsyncMcts<G>` (src/async_mcts.rs:14-115), whose callers are
 * `Coach::execute_episode` (src/coach.rs:104-157) and `arena::play_games`
 * (src/arena.rs:62-99).  The entry points below are what a Rust `extern "C"`
 * block would bind to keep `Coach`/`arena` and swap the engine; each cites the
 * reference item it replaces.  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *   - Plain C types only; opaque handles; every output buffer is allocated and
 *     owned by the caller; inputs are borrowed for the duration of the call.
 *   - Every call returns an az_status (0 = ok).  The reference panics instead
 *     (unwrap/assert!); the panic sites map onto the status codes below.
 *   - A handle is used by one host thread at a time; calls are synchronous on
 *     return.  One HIP stream per engine.  Engines on different devices may be
 *     driven from different host threads freely.  Two engines on the SAME device
 *     driven from two threads at once need "search_graph" 0 on both: while one
 *     thread captures its search loop as a hipGraph, HIP refuses the blocking
 *     copies of the other thread (hipErrorStreamCaptureImplicit, reported as
 *     AZ_ERR_HIP).  Measured gain of that arrangement: none (profiles/README.md).
 *     az_arena overlaps its two models' searches on two HIP streams; streams share the device's few hardware queues, so a
 *     process that keeps many other streams alive can make the two share one (measured: 2755 -> 1860 games/s with one extra
 *     idle stream).
 *   - Game state: Connect Four as two 7x6 bitboards in canonical form
 *     {mine, theirs} (side to move = mine); bit(col,row) = col*7 + row with
 *     row 0 = bottom; bit col*7+6 is always clear.
 *   - Feature tensors are NCHW [B,2,6,7] f32, plane 0 = side to move, plane 1 =
 *     opponent, row 0 = top (connect_four_game.rs:219-237 with repair S8).
 *   - Pointers may be host or device memory unless stated otherwise (the
 *     library copies with hipMemcpyDefault).
 */
#ifndef AZ_ENGINE_H
#define AZ_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_ACTIONS 7       /* connect_four_game.rs:14 */
#define AZ_FEATURES 84     /* 2*6*7, connect_four_game.rs:86-88 */
#define AZ_MAX_PLIES 42

typedef enum az_status {
    AZ_OK = 0,
    AZ_ERR_BAD_ARGUMENT = 1,   /* contract violations the reference asserts (src/async_mcts.rs:192, src/coach.rs:83) */
    AZ_ERR_CAPACITY = 2,       /* node arena exhausted: assert!(idx < buf.len()), src/node.rs:237 */
    AZ_ERR_HIP = 3,            /* HIP runtime / launch failure */
    AZ_ERR_INVALID_MOVE = 4,   /* arena validity assert, src/arena.rs:31-35 */
    AZ_ERR_TERMINAL_ROOT = 5,  /* get_action_prob on a finished game: p.unwrap() panic, src/async_mcts.rs:85 */
    AZ_ERR_NO_MODEL = 6,       /* model_id was never initialised / loaded */
    AZ_ERR_IO = 7,             /* checkpoint read/write */
    AZ_ERR_UNSUPPORTED = 8
} az_status;

/* Which network answers NNet::predict for a model id. */
typedef enum az_net_kind {
    AZ_NET_STUB = 0,   /* DumbConnectFourNnet, examples/connect_four.rs:12-43: pi = 1/7, v = +1 */
    AZ_NET_HASH = 1,   /* deterministic pseudo-random net (test fixture, exact in f32) */
    AZ_NET_CONV = 2    /* policy+value conv net, connect_four_net.py:20-95, bf16 MFMA */
} az_net_kind;

typedef struct az_engine az_engine;
typedef struct az_tree az_tree;

typedef struct az_config {
    int32_t device;        /* HIP device ordinal */
    int32_t max_batch;     /* largest leaf batch the conv net is sized for (0 = 8192) */
    int32_t net_channels;  /* conv width; 0 = 512 (connect_four_net.py:21) */
    int32_t profile;       /* !=0: bracket the dominant kernels with HIP events (az_get_stats) */
    int32_t game;          /* az_game: which Game (src/game.rs:10-28) the trees play; 0 = Connect Four */
} az_config;

/* The Game the engine's trees are instantiated for (trait Game, src/game.rs:10-28; the device-side seam is
 * alphazero-rs_amd/csrc/az_game.h).  AZ_GAME_CONNECT_FOUR is the reference's one implementor
 * (examples/connect_four_lib/connect_four_game.rs); AZ_GAME_CONNECT_THREE is the same board, moves, features and nets with
 * three in a row winning: the seam's second instantiation (its CPU twin lives with the tests), there to prove the rules are a policy. */
typedef enum az_game { AZ_GAME_CONNECT_FOUR = 0, AZ_GAME_CONNECT_THREE = 1 } az_game;

/* Counters (SURVEY.md 8b "Introspection"); all cumulative since az_create / az_reset_stats. */
typedef struct az_stats {
    uint64_t games;          /* finished self-play / arena games */
    uint64_t moves;          /* get_action_prob calls */
    uint64_t simulations;    /* search_iteration calls, src/async_mcts.rs:219 */
    uint64_t expansions;     /* upgrade -> Some(true), src/node.rs:290-323 */
    uint64_t leaf_evals;     /* NNet::predict rows (root priors included) */
    uint64_t link_hits;      /* upgrade -> Some(false), src/node.rs:285-289 */
    uint64_t terminal_hits;  /* simulations that ended on an existing terminal node */
    uint64_t depth_sum;      /* best_child calls (selection levels) */
    uint64_t samples;        /* training tuples emitted (before symmetries) */
    uint64_t net_launches;   /* conv2 launches timed (profile mode) */
    double net_conv2_ms;     /* summed conv2 kernel time (profile mode) */
    double net_conv2_flops;  /* summed conv2 MFMA flops (profile mode); 0 while conv2 runs as a table lookup */
    double net_total_ms;     /* summed whole-forward time (profile mode) */
    double net_total_flops;  /* summed whole-forward flops of the layers that ran as arithmetic (profile mode) */
    double tree_ms;          /* summed select+backup kernel time (profile mode) */
    double tree_bytes;       /* summed algorithmic tree bytes, SURVEY.md 8d (profile mode) */
    double device_ms;        /* summed wall time spent inside engine calls */
    /* leaf de-duplication ("eval_dedup"): requested = rows the trees asked for (== leaf_evals), executed = rows the net
     * really ran (requested - cache hits - in-batch duplicates; == requested when de-duplication is off) */
    uint64_t leaf_rows_requested;
    uint64_t leaf_rows_executed;
    uint64_t eval_cache_hits;
    uint64_t eval_batch_dups;
    uint64_t eval_cache_inserts;
    double net_conv3_ms;           /* summed conv3 kernel time (profile mode) */
    double net_conv3_flops;        /* summed conv3 algorithmic flops (profile mode) */
    double net_conv2_bytes;        /* conv2 as a table ("conv2_table"): summed algorithmic bytes (table rows gathered + rows written) */
    uint64_t tree_launches;        /* select/backup launches (profile mode brackets every "profile_every"-th of them) */
    uint64_t tree_launches_timed;  /* ... of which tree_ms was measured on */
    uint64_t tree_arena_allocs;  /* tree arenas hipMalloc'ed by az_selfplay / az_arena since az_create (kept and reused across
                                  * calls of the same shape; not cleared by az_reset_stats) */
    double net_conv4_ms;           /* summed conv4 kernel time (profile mode) */
    double net_conv4_flops;
    double net_fc_ms;              /* summed fc1 + fc2 + heads kernel time (profile mode) */
    double net_fc_flops;
    double net_rows_timed;         /* executed rows of the timed forwards (profile mode) */
    uint64_t abandoned_sims;     /* num_threads > 1 only: simulations abandoned where the reference has no legal continuation
                                  * (every child Locked, src/node.rs:366-367; a link into a Locked node, :354); counted in `simulations` */
    uint64_t net_conv3_image_rows;      /* rows (boards) conv3 processed on the image-resident kernel k_conv3_auto, counted on the device ... */
    uint64_t net_conv3_image_launches;  /* ... and the launches of it that did that work (a launch that finds the batch on the small-batch
                                         * kernel's side of the hand-over exits at once and is not counted): what a profiler's average
                                         * duration of k_conv3_auto has to be divided into.  Always on (two atomics per launch). */
} az_stats;

/* ---- lifecycle ---------------------------------------------------------- */
az_status az_create(const az_config* cfg, az_engine** out);
/* Destroy every az_tree of the engine first: a tree borrows the engine's stream. */
void az_destroy(az_engine* e);
const char* az_last_error(const az_engine* e);
/* Tuning switches (no reference counterpart).  EVERY option is state of the engine it is set on: two engines in one process never
 * see each other's settings.  Unknown keys or values return AZ_ERR_BAD_ARGUMENT.  Keys of libaz_engine.so (each choice of a
 * bit-identical group computes the same bits; tests/test_net_gpu.py):
 *   the net  "conv2_table"  1 (default): conv1 + conv2 as nine gathered rows of a per-model table (conv2 is linear in conv1's output,
 *                           which is one of 3^9 table rows per position; 198 of the net's 329 MFLOP per leaf are never executed);
 *                           0: conv2 as the MFMA implicit GEMM -- the same function with its own rounding (each batch-independent and
 *                           within the stated tolerance of the fp32 reference)
 *            "conv3_small"  1 (default): conv3 of a small expected batch runs on the 4-stage LDS-DMA ring; 0: never.  Bit-identical
 *            "conv3_tail"   1 (default): a short last round of conv3 workgroups is cut into half tiles; 0: full tiles.  Bit-identical
 *            "conv3_planes" 1 (default): conv3's LDS image in the bank-conflict-free layout; 0: image rows in order.  Bit-identical
 *            "ring_packed"  1 (default): the LDS-DMA ring kernels (conv4, fc1, fc2, small conv3) stream their weight stages from a packed
 *                           copy of the model (16 KiB of consecutive bytes per stage); 0: from the [N][K] weights.  Bit-identical
 *            "narrow_rows"  n (default 32, 0 = off): batches of at most n boards (conv3; 2n for conv4, 4n for the FCs) run the
 *                           register-fed skinny GEMM; the hand-over is decided on the device from the exact row count.  Bit-identical
 *   search   "search_graph" n (default 20, even, 0 = off): n simulation steps per captured hipGraph replay (conv nets) ...
 *            "search_graph_rows" n (default 1024): ... for searches whose expected leaf batch has at most n rows (the arena, the drain
 *                           of a self-play call, single trees: there the host's launch calls set the pace; on big batches the kernels do)
 *            "fused_search" 1 (default): the stub / hash nets run a whole search in one launch; 0: one launch per simulation
 *            "selfplay_async" 0 (default) / 1: az_selfplay with FREE-RUNNING slots -- every slot runs backup, move, next root and select on
 *                           its own timeline inside the tree kernel, "selfplay_async_launches" (default 2) launches share one leaf batch
 *                           (trees the cache answered go on, trees that took a row wait for the forward), at most
 *                           "selfplay_async_iters" (default 6) stages per slot and launch.  Same games bit for bit (a game depends on
 *                           its seed, its id and the net's rows, never on the schedule); fewer, larger forwards
 *            "tree_block4"  1 (default): four waves per workgroup in the select / backup kernel; 0: one
 *   leaf de-duplication (bit-exact: a row's (pi, v) depends on its state alone; the reference's per-tree analogue is `seen`,
 *   src/node.rs:282-289)
 *            "eval_dedup"   0 off / 1 conv nets (default) / 2 every net: each distinct state of a leaf batch is evaluated once
 *            "eval_cache_log2"  upper bound of log2 entries of the engine's evaluation cache (default 30, 0 = none, 10..30; 40 bytes per
 *                           entry).  A call (or session) allocates and clears only what ITS games can fill (4 x its bound on
 *                           inserted rows, at least 2^10): a 1-tree, 25-simulation call touches 40 KB, a call of 8192 episodes
 *                           11 GB, and only bench-sized ones (65536 episodes and more) the full 43 GB -- a seventh of the
 *                           device's 288 GB, and worth it: a self-play session of 1.6 M episodes runs at 10.6 k games/s with
 *                           2^30 entries and at 7.9 k with 2^27, whose table is full after a fifth of it
 *            "eval_cache_max_stones"  only states with at most that many stones are cached (default 42)
 *            "eval_cache_persist"  0 (default): every az_selfplay / az_arena / az_tree_get_action_prob call starts from an empty cache;
 *                           1: entries live until the model's weights change (the full "eval_cache_log2" table)
 *            "dedup_stats"  1 (default) / 0: maintain the five leaf-row counters of az_stats
 *   profile  "profile"      0 / 1: the HIP-event brackets of az_config.profile, switched between calls (bracketed searches launch every
 *                           kernel on its own; a timed region runs with them off, a separate pass with them on gives the kernel times)
 *            "profile_every" n (default 1): with the brackets on, bracket every n-th simulation step (the net_* and tree_ms sums
 *                           then cover that sample of launches; a bracket costs a little idle time between kernels)
 *   NNet::train  "train_epochs" (10), "train_batch" (64, <= 256), "train_seed" (0), "train_lr_e9" (1000000 = 1e-3),
 *            "train_dropout_e6" (300000 = 0.3), "train_graph" 1 (default) / 0: replay a step's launches as a captured hipGraph,
 *            "train_gemm" 1 (default): dgrad / wgrad as bf16 x 3 on the bf16 matrix cores (gradients within 1e-5 of float64
 *                           autograd), 0: every GEMM on v_mfma_f32_16x16x4_f32 (1e-6).  Two numerics classes: the trained
 *                           weights differ, as they do between two f32 summation orders (parity of NNet::train is unpinned by
 *                           the reference, whose training script cannot run; tests/test_train_gpu.py bounds the drift),
 *            "train_fwd_x3" 1 (default): the forward GEMMs of conv2..conv4 as f16 x 3 on the f16 matrix cores (activations x 64 and
 *                           weights x 256 split into half-precision hi / lo pairs, three products, f32 accumulate: 2^-22, the grade of
 *                           the f32 kernel's own accumulation -- gradients stay within 1e-5 of float64 autograd), 0: on
 *                           v_mfma_f32_16x16x4_f32; "train_gemm3_ring" 1 (default) / 0: the big x 3 GEMMs on the 256 x 128 ring kernel,
 *            "train_wgrad_tr" 1 (default) / 0: conv wgrad from the operands as stored, transposed LDS reads (k_wgrad3_tr) instead of
 *                           transpose kernels + k_gemm3; "train_implicit" 1 (default) / 0: conv2..conv4's GEMMs gather their A rows from
 *                           the activations instead of reading im2col matrices (needs batch % 16 == 0 and net_channels % 256 == 0,
 *                           else the im2col path runs); every combination is held to the same bars by tests/test_train_gpu.py,
 *            "train_fork" 0 (default) / 1: with "train_gemm" 1, a step's weight split and wgrad chains run on a second stream
 *                           branch beside the BatchNorm-backward / dgrad chain (bit-identical; measured no faster, so off),
 *            "train_fwd_dma" 1 (default): the forward GEMMs' tiles go global -> LDS by LDS-DMA (k_gemm_f32_dma), 0: register-staged
 * libaz_engine_diag.so (the same sources built with -DAZ_DIAG; alphazero-rs_amd/build.py) additionally takes the keys of the
 * SUPERSEDED kernel generations and the TIMING ABLATIONS WITH WRONG RESULTS -- "gemm_variant", "fc_ring", "ring_tile", "conv3_ring",
 * "conv2_pipe", "conv3_pipe", "conv1_table", "conv4_big", "conv2_table" = 2, "tree_stamps", "print_*" -- which the shipped library
 * refuses (it accepts their default values, so a host may set them unconditionally); csrc/az_net.hip, csrc/az_net_diag.inc. */
az_status az_set_option(az_engine* e, const char* key, int64_t value);
az_status az_get_stats(az_engine* e, az_stats* out);
az_status az_reset_stats(az_engine* e);

/* ---- NNet trait, src/nnet.rs:35-45 -------------------------------------- */
/* NNet::new for the stub / hash nets (no weights). salt only matters for AZ_NET_HASH. */
az_status az_net_set_kind(az_engine* e, int32_t model_id, az_net_kind kind, uint64_t salt);
/* Drop a model id and its device memory (weights + conv1 table, 41 MB at C = 512).  The reference never frees a model:
 * its Python side keeps one checkpoint per id on disk (src/nnet.rs:36); a long Coach::learn run (src/coach.rs:296-390: a new
 * id per accepted iteration) calls this for the superseded id.  The activation workspace is per stream, not per model. */
az_status az_net_free(az_engine* e, int32_t model_id);
/* NNet::new with random init: Glorot-uniform kernels, zero bias, BN gamma=1 beta=0 mean=0 var=1 eps=1e-3. */
az_status az_net_init_random(az_engine* e, int32_t model_id, uint64_t seed);
/* NNet::new(checkpoint) / save: flat f32 file, layout in DESIGN.md "weights file". */
az_status az_net_load(az_engine* e, int32_t model_id, const char* path);
az_status az_net_save(az_engine* e, int32_t model_id, const char* path);
/* Raw f32 parameter exchange (same order as the weights file); count from az_net_param_count. */
int64_t az_net_param_count(const az_engine* e);
az_status az_net_set_params(az_engine* e, int32_t model_id, const float* params, int64_t n);
az_status az_net_get_params(az_engine* e, int32_t model_id, float* params, int64_t n);
/* NNet::predict(board [B,2,6,7], model_id) -> (pi [B,7], v [B]), src/nnet.rs:40-44 */
az_status az_net_predict(az_engine* e, int32_t model_id, const float* boards, int32_t B, float* pi, float* v);
/* Same on canonical bitboards [B,2] (what the search feeds the net). */
az_status az_net_predict_states(az_engine* e, int32_t model_id, const uint64_t* states, int32_t B, float* pi, float* v);
/* NNet::train(examples, previous_model_id, model_id), src/nnet.rs:38: start from the weights of prev_id, run the
 * reference's recipe on the device (connect_four_net.py:13-21, :102-151: loss = softmax cross-entropy(pi) + mean
 * squared error(v), Adam, BatchNorm in training mode, dropout on the two FC layers; epochs x (n / batch) steps on
 * batches drawn with replacement), store the result under id. boards [n,2,6,7], pis [n,7], vs [n] f32, host or
 * device. f32 parameters, activations, gradients and optimiser state; the forward GEMMs on the f32 matrix cores, dgrad / wgrad as
 * bf16 x 3 with f32 accumulation ("train_gemm"; csrc/az_train.hip). Hyper-parameters through
 * az_set_option: "train_epochs" (10), "train_batch" (64, <= 256), "train_seed" (0), "train_lr_e9" (1000000 = 1e-3),
 * "train_dropout_e6" (300000 = 0.3). Batches and dropout masks come from the build's counter RNG (B7). */
az_status az_net_train(az_engine* e, int32_t prev_id, int32_t id, const float* boards, const float* pis,
                       const float* vs, int64_t n);
/* (loss_pi, loss_v) averaged over each epoch of the last az_net_train: writes min(epochs, cap_epochs) pairs to out
 * (may be NULL) and returns the number of epochs. */
int32_t az_net_train_history(const az_engine* e, float* out, int32_t cap_epochs);
/* Fine-grained parity entries (what az_net_train is made of). begin: load prev_id's weights, zero the Adam moments.
 * step: ONE optimisation step on an explicit batch (2 <= b <= 256); mask_seed keys this step's dropout masks;
 * apply = 0 leaves the weights alone (loss and gradients only, BatchNorm moving averages still advance);
 * loss_out[2] = (loss_pi, loss_v), grads_out[az_net_param_count] = d loss / d parameter in weights-file order
 * (both may be NULL). end: store the weights under model_id. */
az_status az_net_train_begin(az_engine* e, int32_t prev_id);
az_status az_net_train_step(az_engine* e, const float* boards, const float* pis, const float* vs, int32_t b,
                            uint64_t mask_seed, int32_t apply, float* loss_out, float* grads_out);
az_status az_net_train_end(az_engine* e, int32_t model_id);

/* ---- AsyncMcts, src/async_mcts.rs:14-115 -------------------------------- */
/* n_games independent AsyncMcts::default(reserve, num_sims, num_threads, max_depth, model_id, cpuct, ..)
 * (src/async_mcts.rs:27-48), each rooted at the initial board (NodeStore::new, src/node.rs:156-166).
 * num_threads = simulations in flight per tree (src/async_mcts.rs:191-217; num_sims % num_threads == 0, :192).
 * 1 is the reference's only deterministic mode.  num_threads > 1 runs the reference's tree-parallel search as a
 * deterministic LOCK-STEP schedule (one legal execution of the racy original; DESIGN.md "several simulations in flight"):
 * per step the threads select in thread order, each seeing the visits and virtual losses (src/node.rs:77-80) of the earlier
 * ones and the `Locked` filter of src/node.rs:359-365 on leaves they hold; the step's leaves are evaluated together and
 * backed up in thread order.  At most 8 threads. */
az_status az_tree_create(az_engine* e, int32_t n_games, uint64_t reserve, int32_t num_sims, int32_t num_threads,
                         int32_t max_depth, int32_t model_id, int32_t cpuct, az_tree** out);
void az_tree_destroy(az_tree* t);
/* AsyncMcts::from_state(s, ..) (src/async_mcts.rs:50-72, NodeStore::from_root, src/node.rs:168-177): forget every
 * tree of the batch and root tree g at root_states[g] (canonical bitboards [G,2]); NULL = the initial board. */
az_status az_tree_reset(az_tree* t, const uint64_t* root_states);
/* get_action_prob(&self, s, temp, episode_id, rng) for every tree at once (src/async_mcts.rs:74-115).
 * states [G,2]; outputs pi [G,7], counts [G,7] (child N), q [G,7] (child Q); counts/q may be NULL.
 * RNG (temp == 0 tie-break) = stream (seed, first_game_id + g, ply = stones on board). */
az_status az_tree_get_action_prob(az_tree* t, const uint64_t* states, float temp, uint64_t seed,
                                  uint64_t first_game_id, float* pi, uint16_t* counts, float* q);
/* Record every NNet::predict the search issues, per tree, in order (replay parity). cap = records per tree. */
az_status az_tree_record_evals(az_tree* t, int32_t cap);
/* Copy out the record log: rec_count [G]; states [G,cap,2], pis [G,cap,7], vs [G,cap] (any may be NULL). */
az_status az_tree_get_evals(az_tree* t, int32_t* rec_count, uint64_t* states, float* pis, float* vs);
/* Node count (NodeStore::len, src/node.rs:372-374) per tree, [G]. */
az_status az_tree_node_counts(az_tree* t, uint32_t* out);

/* ---- Coach::execute_episode x many, src/coach.rs:104-157 ------------------ */
typedef struct az_selfplay_params {
    int32_t n_games;         /* episodes to play in this call (global ids first_game_id .. +n_games) */
    int32_t concurrent;      /* game slots resident at once (0 = n_games); finished slots are refilled */
    int32_t num_sims;        /* src/coach.rs:30 */
    int32_t temp_threshold;  /* src/coach.rs:22 */
    int32_t max_depth;       /* src/coach.rs:32 */
    int32_t cpuct;           /* src/coach.rs:33 */
    int32_t model_id;
    int32_t symmetries;      /* !=0: emit identity + mirror per position (get_symmetries), else identity only */
    uint64_t reserve;        /* mcts_reserve_size, src/coach.rs:20 (clamped to the reachable bound) */
    uint64_t seed;
    uint64_t first_game_id;
    int32_t record_evals;    /* records per EPISODE kept for az_selfplay_get_evals (0 = off); works with slot refill */
    int32_t num_sim_threads; /* simulations in flight per tree, src/coach.rs:31, :249 (0 = 1; see az_tree_create) */
} az_selfplay_params;

/* Training tuples (s, pi, z) in game-id order then ply order; TrainingSample, src/nnet.rs:22-27. */
typedef struct az_samples {
    int64_t capacity;     /* in: tuples the arrays can hold (n_games*42, x2 with symmetries, always suffices) */
    int64_t count;        /* out */
    uint64_t* states;     /* [capacity,2] canonical bitboards (may be NULL) */
    float* boards;        /* [capacity,2,6,7] features (may be NULL) */
    float* pis;           /* [capacity,7] */
    float* zs;            /* [capacity] */
    int32_t* game_len;    /* [n_games] plies per game (may be NULL) */
    uint8_t* moves;       /* [n_games,42] actions played (may be NULL) */
} az_samples;

az_status az_selfplay(az_engine* e, const az_selfplay_params* p, az_samples* out);
/* The same as a SESSION (no reference counterpart: the reference's Coach collects an iteration's episodes from a rayon pool,
 * src/coach.rs:246-260, in whatever order they finish).  az_selfplay_begin fixes the episodes (p->n_games of them, ids 0 ..
 * n_games-1, first_game_id / seed as in az_selfplay) and fills the slots; az_selfplay_next(k, out) plays until the NEXT k episodes
 * in id order have finished and returns exactly the tuples az_selfplay would return for them (an episode depends on its id,
 * the seed and the net alone) -- while the slots they freed already play later episodes.  A host that fetches its episodes in
 * chunks (one training-set shard, one bench step at a time) thereby pays the drain of the last slots once per session, not
 * once per chunk.  az_selfplay_end closes the session (az_destroy does too); one session per engine; the model may not be
 * changed while it is open.  az_selfplay is begin + next(n_games) + end. */
az_status az_selfplay_begin(az_engine* e, const az_selfplay_params* p);
az_status az_selfplay_next(az_engine* e, int32_t n_games, az_samples* out);
az_status az_selfplay_end(az_engine* e);
/* Eval log of the last az_selfplay with record_evals > 0: rec_count [n_games], states [n_games,cap,2], ... */
az_status az_selfplay_get_evals(az_engine* e, int32_t* rec_count, uint64_t* states, float* pis, float* vs);

/* ---- arena::play_games, src/arena.rs:62-99 + gate, src/coach.rs:377-390 ---- */
typedef struct az_arena_params {
    int32_t num_games;      /* num/2 per seating, src/arena.rs:83 */
    int32_t num_sims;
    int32_t max_depth;
    int32_t cpuct;
    int32_t new_model_id;   /* first listed player ("new", nmcts, src/coach.rs:345-354) */
    int32_t old_model_id;   /* second listed player ("old", pmcts, src/coach.rs:333-343) */
    uint64_t reserve;
    uint64_t seed;
    /* Sharding (one process per GPU): this call plays games [first_game, first_game + num_games) of an arena of
     * total_games (0 = num_games, first_game must then be 0).  Seating (game < total/2 -> (new, old)) and the RNG
     * stream use the GLOBAL game index, so shards add up to exactly the unsharded arena (3-counter all-reduce).
     * A sharded call plays all num_games of its range (the caller splits an even total). */
    int32_t first_game;
    int32_t total_games;
    int32_t record_evals;     /* records per game and player kept for az_arena_get_evals (0 = off) */
    int32_t num_sim_threads;  /* simulations in flight per tree, src/coach.rs:340, :351 (0 = 1; see az_tree_create) */
    /* play_games' `board: Option<G>` (src/arena.rs:62-67, :12-16): every game starts from this position with the first
     * seat (cur_player = +1) to move; start_board = {first seat's stones, second seat's stones}.  use_start_board = 0: the
     * initial board (None). */
    int32_t use_start_board;
    /* sharded call (total_games > 0) on an engine with a communicator (az_comm_init): != 0 sums out_wld over the ranks
     * (one 3-counter all-reduce over RCCL), so every rank returns the whole arena's tally */
    int32_t allreduce_wld;
    uint64_t start_board[2];
} az_arena_params;
/* out_wld[3] = {Win, Loss, Draw} for the new model (GameResult, src/arena.rs:54-59);
 * results [num_games] (may be NULL): +1 first seat won, -1 second seat won, 0 draw (play_game, src/arena.rs:51). */
az_status az_arena(az_engine* e, const az_arena_params* p, uint64_t out_wld[3], int8_t* results);
/* Eval log of the last az_arena with record_evals > 0, for the trees of player `which` (0 = new model, 1 = old model):
 * rec_count [num_games], states [num_games,cap,2], pis [num_games,cap,7], vs [num_games,cap] (any may be NULL). */
az_status az_arena_get_evals(az_engine* e, int32_t which, int32_t* rec_count, uint64_t* states, float* pis, float* vs);
/* Move record of the last az_arena: game_len [num_games] plies played, moves [num_games][AZ_MAX_PLIES] the actions in order (the
 * board sequence play_game's `verbose` prints, src/arena.rs:20-27; what one reaches for when an arena game diverges).  Either
 * may be NULL. */
az_status az_arena_get_moves(az_engine* e, int32_t* game_len, uint8_t* moves);

/* ---- the collective of the sharded Coach loop (no reference counterpart: the reference is one process,
 * src/coach.rs:241-272 fans episodes out over a rayon pool; here one process per GPU plays a shard of the global
 * episode ids and the (s, pi, z) tuples meet once per episode batch) --------------------------------------------------
 * RCCL over xGMI on the engine's own stream.  az_comm_unique_id is called on ONE rank; the host ships the 128 bytes to the
 * others by its own means (a file, MPI, torch.distributed ...) and every rank calls az_comm_init with them. */
#define AZ_COMM_ID_BYTES 128
az_status az_comm_unique_id(az_engine* e, uint8_t id[AZ_COMM_ID_BYTES]);
az_status az_comm_init(az_engine* e, int32_t rank, int32_t world, const uint8_t id[AZ_COMM_ID_BYTES]);
az_status az_comm_destroy(az_engine* e);
/* One gather of the packed tuples of `local` (count tuples: states [count,2], pis [count,7], zs [count]; host or device)
 * to dst_rank: an all-gather of the per-rank counts (counts_out [world], may be NULL), then ONE gather of 48-byte packed
 * tuples in rank order.  On dst_rank `gathered` receives them (capacity in, count out; states / pis / zs required, host or
 * device); on other ranks `gathered` may be NULL.  dst_rank = -1: every rank receives (the same exchange as an all-gather: the
 * replicated trainer of the sharded Coach loop).  Collective: every rank of the communicator calls it. */
az_status az_gather_samples(az_engine* e, const az_samples* local, int32_t dst_rank, az_samples* gathered, int64_t* counts_out);
/* In-place sum over the ranks of n (<= 64) u64 counters (the arena's W/L/D; host memory).  Collective. */
az_status az_allreduce_u64(az_engine* e, uint64_t* values, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* AZ_ENGINE_H */
