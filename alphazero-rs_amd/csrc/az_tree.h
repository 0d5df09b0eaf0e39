// az_tree.h -- device-side layout of a batch of MCTS trees and the kernel launchers.
//
// One TreeDev = G independent NodeStores (src/node.rs:129-375), G*R node slots in HBM, a game's slots contiguous.
//
// Node record = 16 bytes (everything a selection level needs from a child):
//     +0  u64 ctr    packed win counter 0xWWWWWWWW_NNNN_VVVV (src/node.rs:17)
//     +8  u32 prior  f32 bits of the edge prior parent -> this slot (parent.mu.p[a], src/node.rs:354)
//     +12 u32 word   a | nchild << 3 | ecode << 6 | has_prior << 8 | locked << 9 | kind << 10 | payload << 12
//                    kind 0: placeholder; 1: link, payload = tree-local slot of the canonical node (NodeLink.1, src/node.rs:129);
//                    2: expanded (mu.s is Some), payload = index of its child block.  20 payload bits: a tree holds at most 2^20 slots
//                    (the reference example's reserve_space of 1,000,000 fits)
// plus, for expanded nodes only, key[slot] = Game::pack(state) (NodeMutableState.s) in an array of its own: the search reads it once
// per simulation (the expansion plays the parent's state) and the transposition probe compares it.
// Children are pushed contiguously at expansion (src/node.rs:313-317) into a CHILD BLOCK of Game::GROUP (8) slots = ONE 128-byte
// line: lane j of the 8 lanes that serve a game loads child j's record, so a selection level costs exactly one line (round 3's
// 32-byte records, with the state in the record, cost two; round 1's separate rec / ctr / state arrays four to six), and the
// record of the chosen child is the next level's parent.  Per-tree search state is one 64-byte TreeHead.
#pragma once
#include "az_common.h"
#include "az_game.h"

namespace az {

enum StatIdx { ST_SIMS = 0, ST_EXPANSIONS, ST_LEAF_EVALS, ST_LINK_HITS, ST_TERMINAL_HITS, ST_DEPTH_SUM, ST_COUNT };
constexpr int ST_ABANDONED = ST_COUNT;      // k_harvest's 7th total: simulations abandoned on an all-Locked child set / a Locked link target
constexpr int ST_TOTALS = ST_COUNT + 1;     // (several simulations in flight only; counted in the per-thread lines, not in TreeHead.stat)
constexpr int MAX_SIM_THREADS = 8;
enum ErrIdx { ERR_CAPACITY = 0, ERR_TERMINAL_ROOT = 1, ERR_PATH = 2, ERR_HASH_FULL = 3, ERR_COUNT = 4 };
// LEAF_ROOT: the root's own evaluation (S1 / S10: no simulation, its backup only stores the prior)
enum LeafKind { LEAF_NONE = 0, LEAF_VALUE = 1, LEAF_EVAL = 2, LEAF_ROOT = 3 };

constexpr int BLOCK_SLOTS = 8;         // slots of a child block (== Game::GROUP)
constexpr uint32_t MAX_TREE_SLOTS = 1u << 20;   // links and child blocks are 20-bit fields of the node record

// per-tree search state: one 64-byte line, read at the start and written at the end of every tree kernel
struct TreeHead {
    uint32_t len;        // bump pointer in slots (a multiple of BLOCK_SLOTS)
    uint32_t count;      // NodeStore::len (src/node.rs:134, :372-374): nodes pushed (root + placeholders)
    uint32_t root;       // root node of the current get_action_prob
    uint32_t active;     // tree takes part in the current search
    uint32_t leaf;       // node the simulation stopped on
    uint32_t leaf_kind;  // LeafKind
    float leaf_val;      // value when LEAF_VALUE
    uint32_t src;        // where the leaf's (pi, v) comes from (SRC_*), LEAF_EVAL
    uint32_t path_len;   // node_path (src/async_mcts.rs:229) entries in TreeDev.path
    uint32_t log_len;    // eval-log records written
    uint32_t stat[ST_COUNT];   // counters since the last harvest
};
static_assert(sizeof(TreeHead) == 64, "TreeHead is one 64-byte line");
// What a tree keeps per launch is one 128-byte line: the head and the first 16 entries of node_path (src/async_mcts.rs:229);
// lane j of the tree's 8 lanes holds entries j and j + 8 in registers, so pushing and walking the path costs no memory round
// trip (longer paths spill to TreeDev.path).
constexpr int PATH_INLINE = 16;
struct TreeLine {
    TreeHead head;
    uint32_t path16[PATH_INLINE];
};
static_assert(sizeof(TreeLine) == 128, "head + inline path = one 128-byte line");

struct TreeDev {
    int32_t G;               // trees
    uint32_t R;              // slots per tree (a multiple of BLOCK_SLOTS)
    uint32_t H;              // hash entries per tree (power of two)
    int32_t game;            // az_game: which Game policy the kernels are instantiated with (0 = ConnectFour)
    uint32_t reserve_nodes;  // reserve_space (src/node.rs:146): pushes beyond it are the reference's assert (src/node.rs:237)
    uint4* node;             // [G*R] 16-byte records {ctr.lo, ctr.hi, prior, word}
    unsigned long long* key; // [G*R] Game::pack(state) of the expanded nodes
    uint32_t* hash;          // [G*H] `seen` (src/node.rs:135): open-addressing table of node slots, key = the node's own key word
    TreeLine* head;          // [G]
    uint32_t* path;          // [G*T*PATH_CAP] node_path entries beyond PATH_INLINE
    // Several simulations in flight per tree (num_threads = T > 1, src/async_mcts.rs:191-217): the per-simulation part of the search
    // state (leaf, leaf_kind, leaf_val, src, path_len + the inline path; stat[0] counts abandoned simulations) of thread tt of tree g
    // is line g*T + tt; the TreeHead of t.head keeps what the tree's threads share (len, count, root, active, log_len, stat).
    int32_t block4;          // k_backup_select as 4-wave workgroups (one row-counter atomic per workgroup); 0 = one wave per workgroup ("tree_block4")
    int32_t T;               // 1 = the single-simulation kernels
    TreeLine* thr;           // [G*T] when T > 1, else nullptr
    uint32_t* err;           // [ERR_COUNT]
    // eval log (replay parity): raw (pi, v) of every NNet::predict row, per tree, in order
    int32_t log_cap;
    ulonglong2* log_state;   // [G*log_cap]
    float* log_pi;           // [G*log_cap*7]
    float* log_v;            // [G*log_cap]
    const int32_t* log_row;  // [G] or nullptr: log row of tree g (az_selfplay: the slot's current episode, so a log survives slot refills); nullptr = g
};

// Leaf batch handed to the net (src/async_mcts.rs:117-189 restated as lanes): the DISTINCT states the trees of one
// simulation step need evaluated.  A tree whose leaf goes to the net either takes the next row (one atomicAdd per wave of 8
// trees) or -- with de-duplication -- finds that another tree already did, or that the engine's evaluation cache holds
// the state; its TreeHead.src says where its (pi, v) will be.  The row ORDER of a batch is scheduling-dependent, the results
// are not: a row's (pi, v) depends on nothing but its state (tests/test_net_gpu.py).
struct EvalBatch {
    int32_t cap;
    uint32_t* n;          // device: rows in this batch
    ulonglong2* state;    // [cap] canonical state to featurise (to_features, connect_four_game.rs:219-237)
    float* pi;            // [cap*8] net output: pi[0..6], v in slot 7
    float* v;             // [cap]
    uint32_t* max_n;      // device, may be nullptr: largest row count of any batch so far (feeds the host's tile choice)
    // ---- de-duplication (dedup == 0: every requesting tree takes its own row) --------------------------------------
    // Thousands of games share their openings and a row's (pi, v) depends on its state alone (BatchNorm is folded, every
    // row's K-sum has one order), so evaluating a state once per batch -- or once per call, through the engine's evaluation
    // cache -- is bit-exact.  The per-tree analogue in the reference is `seen` (src/node.rs:282-289).
    // Election table (open addressing, linear probe): the first tree to CAS its state key in takes a row; later trees with
    // the same key point at the winner's slot.  The launch that CONSUMES a batch (its backups read tuniq, nothing reads tkey any
    // more) clears the batch's keys, so every launch that requests leaves finds its table empty and every launch of a search
    // takes the same arguments -- what lets a run of simulation steps be one hipGraph.
    int32_t dedup;
    unsigned long long* tkey;   // [tmask+1] state key (never 0), 0 = empty
    uint32_t* tuniq;            // [tmask+1] row of the slot's winner
    uint32_t tmask;
};
constexpr uint32_t SRC_TABLE = 0x80000000u, SRC_CACHE = 0x40000000u, SRC_INDEX = 0x3FFFFFFFu;

// Evaluation cache of the engine: state key (49 bits) | model tag (15 bits) -> (pi[7], v), 8-way buckets of one 64-byte key
// line.  Filled by the trees whose row was evaluated (in the backup kernels), looked up when a leaf is requested, never
// evicted; entries of a model die with its tag (a new tag per weight upload).  key == nullptr: off.
// A key can become visible to a lookup of the SAME launch that publishes it; its payload is read one launch later.
struct EvalCache {
    unsigned long long* key;   // [(bmask+1)*8], 0 = empty
    float* pv;                 // [(bmask+1)*8][8]
    uint32_t bmask;            // buckets - 1
    uint32_t max_stones;       // only states with at most this many stones are inserted
    unsigned long long tag;    // model tag, already shifted to bits 49..63
    unsigned long long* stat;  // [DD_REPLICAS][DD_STRIDE]: every workgroup adds to replica blockIdx % DD_REPLICAS (its own 64-byte line)
};
enum DedupStat { DD_REQUESTED = 0, DD_EXECUTED, DD_CACHE_HITS, DD_BATCH_DUPS, DD_INSERTS, DD_COUNT };
// One set of counters is one hot cache line for the ~1000 waves of a launch (their atomics serialise in one L2 channel: 5 % of the
// bench); 256 replicas on lines of their own spread them over the channels, the host sums them once per call.
constexpr int DD_REPLICAS = 256, DD_STRIDE = 8;

struct SearchParams {
    uint32_t max_depth;
    float cpuct_f;        // cpuct is an i32 in the reference (src/async_mcts.rs:21), cast at use (src/node.rs:353)
};

// per-slot self-play state (Coach::execute_episode, src/coach.rs:104-157)
struct GamesDev {
    int32_t C;               // slots (== TreeDev.G)
    ulonglong2* state;       // [C] canonical board of the position to move
    int8_t* player;          // [C] cur_player (+1 / -1)
    int32_t* ply;            // [C] moves played so far
    int32_t* gid;            // [C] index of the episode within this call, -1 = idle
    uint8_t* need_reset;     // [C] slot was refilled: its tree must be rebuilt
    int32_t* sims;           // [C] free-running self-play (k_async_step): simulations of the current move done, -1 = root not prepared; else nullptr
    // per-episode outputs, indexed by gid
    int32_t n_games;
    ulonglong2* smp_state;   // [n_games*42]
    float* smp_pi;           // [n_games*42*7]
    int8_t* smp_player;      // [n_games*42]
    uint8_t* moves;          // [n_games*42]
    int32_t* g_len;          // [n_games]
    float* g_result;         // [n_games] r = get_game_ended(cur_player) at the end (src/coach.rs:144)
    int8_t* g_final_player;  // [n_games]
    int32_t* g_log_len;      // [n_games] or nullptr: eval-log records of the episode (record_evals)
    // counters: [0] next episode to hand out, [1] episodes finished, [2] active slots
    uint32_t* counters;
};

struct SelfplayMoveParams {
    uint64_t seed;
    uint64_t first_game_id;
    int32_t temp_threshold;
    int32_t refill;          // hand finished slots the next episode
    int32_t done_lo, done_hi; // counters[1] counts the finished episodes with done_lo <= index < done_hi (a session delivers its episodes in chunks)
};

// arena::play_games state (src/arena.rs:7-99): game g < half is seated (new, old), g >= half (old, new)
struct ArenaDev {
    int32_t G, half;       // games in this shard; global index below which a game is seated (new, old)
    int32_t first;         // global index of this shard's game 0
    ulonglong2* state;     // [G] canonical board of the position to move
    int8_t* player;        // [G] cur_player: +1 = first seat to move
    uint8_t* alive;        // [G]
    int8_t* results;       // [G] play_game's return: +1 first seat won, -1 second seat won, 0 draw (src/arena.rs:51)
    uint32_t* counters;    // [0] games still running, [1] invalid-move flag (src/arena.rs:31-35)
    uint8_t* moves;        // [G][42] the actions played, in order (what play_game's `verbose` prints, src/arena.rs:20-27)
    int32_t* len;          // [G] plies played
};

// ---- launchers (all asynchronous on `s`; dispatch on TreeDev.game to the Game policy's instantiation) ------------------
void launch_count_done(const int32_t* g_len, int lo, int hi, uint32_t* counter, hipStream_t s);   // *counter = #{lo <= i < hi : g_len[i] > 0}
void launch_init_heads(const TreeDev& t, hipStream_t s);                      // zero every TreeHead, active = 1
void launch_set_active(const TreeDev& t, uint32_t value, hipStream_t s);
void launch_reset_trees(const TreeDev& t, const uint8_t* flags /*[G] or nullptr = all*/, hipStream_t s,
                        const ulonglong2* roots = nullptr /*[G] root states, nullptr = initial board*/);
void launch_root_prepare(const TreeDev& t, const EvalBatch& eb, const EvalCache& ec, const ulonglong2* root_states, hipStream_t s);
void launch_backup(const TreeDev& t, const EvalBatch& eb, const EvalCache& ec, hipStream_t s);
// backup of simulation i (batch eb_prev) + select of simulation i+1 (leaf requested in eb_next) in one launch
#ifdef AZ_DIAG
bool tree_set_stamps(int on);                                       // diagnostic library: per-wave phase stamps of k_backup_select
bool tree_read_stamps(unsigned long long* out /*[4096 * 8]*/);
#endif
void launch_backup_select(const TreeDev& t, const EvalBatch& eb_prev, const EvalBatch& eb_next, const EvalCache& ec, SearchParams sp,
                          hipStream_t s);
// T > 1: one lock-step STEP of the tree-parallel search: the backups of the previous step's T leaves in thread order (first = 1: the
// root's priors only), then -- unless last -- T selections in thread order, each seeing the earlier ones' visits, virtual losses and locks
void launch_step_mt(const TreeDev& t, const EvalBatch& eb_prev, const EvalBatch& eb_next, const EvalCache& ec, SearchParams sp,
                    int first, int last, hipStream_t s);
// the whole search (root prepare + num_sims simulations + backups) in one launch for the device-function nets
// (kind 0 = stub, 1 = hash fixture): no leaf batch, no kernel boundary per simulation
void launch_search_fixture(const TreeDev& t, const ulonglong2* root_states, SearchParams sp, int num_sims, int kind, uint64_t salt,
                           hipStream_t s);
void launch_root_policy(const TreeDev& t, float temp, uint64_t seed, uint64_t first_game_id, float* pi,
                        uint16_t* counts, float* q, hipStream_t s);
// what one get_action_prob call hands back besides pi / counts / q: written into PINNED host memory by k_call_readback
struct CallReadback {
    unsigned long long totals[ST_TOTALS];
    unsigned long long dd[DD_COUNT];
    uint32_t err[ERR_COUNT];
};
// copies totals (k_harvest's sums), the de-duplication counters' replicas (summed; dd_stat may be nullptr) and the error words to
// out and clears totals and dd_stat: one launch where the host used to issue three blocking copies and two memsets
void launch_call_readback(unsigned long long* totals, unsigned long long* dd_stat, const uint32_t* err, CallReadback* out, hipStream_t s);
// sums the trees' counters into totals[ST_TOTALS] (u64, accumulated) and clears them; node_counts [G] may be nullptr
void launch_harvest(const TreeDev& t, unsigned long long* totals, uint32_t* node_counts, hipStream_t s);
void launch_selfplay_move(const TreeDev& t, const GamesDev& gd, SelfplayMoveParams mp, hipStream_t s);
// free-running self-play: one launch takes every slot through backup / move / root / select until its next leaf needs the net (at most
// max_iters stages); first = the first launch behind a forward (eb_prev holds its rows)
void launch_async_step(const TreeDev& t, const GamesDev& gd, const EvalBatch& eb_prev, const EvalBatch& eb_next, const EvalCache& ec, SearchParams sp,
                       SelfplayMoveParams mp, int num_sims, int first, int max_iters, hipStream_t s);
constexpr int GAME_COUNT = 2;    // 0 = ConnectFour (the reference's Game), 1 = ConnectThree (the seam's second instantiation)
void launch_selfplay_sync_active(const TreeDev& t, const GamesDev& gd, hipStream_t s);
void launch_arena_sync(const TreeDev& t_new, const TreeDev& t_old, const ArenaDev& ad, hipStream_t s);
void launch_arena_move(const TreeDev& t, const ArenaDev& ad, uint64_t seed, hipStream_t s);
void launch_emit_samples(const GamesDev& gd, const int64_t* offsets, int symmetries, ulonglong2* out_states,
                         float* out_boards, float* out_pis, float* out_zs, hipStream_t s);

}  // namespace az
