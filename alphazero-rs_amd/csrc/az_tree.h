// az_tree.h -- device-side layout of a batch of MCTS trees and the kernel launchers.
//
// One TreeDev = G independent NodeStores (src/node.rs:129-375) laid out
// structure-of-arrays in HBM, G*R slots.  A game's slots are contiguous
// ([g*R, (g+1)*R)) so the <=7 children of a node (allocated contiguously,
// src/node.rs:313-317) are one coalesced access by the 8 lanes that serve a game.
#pragma once
#include "az_common.h"

namespace az {

enum StatIdx { ST_SIMS = 0, ST_EXPANSIONS, ST_LEAF_EVALS, ST_LINK_HITS, ST_TERMINAL_HITS, ST_DEPTH_SUM, ST_COUNT };
enum ErrIdx { ERR_CAPACITY = 0, ERR_TERMINAL_ROOT = 1, ERR_PATH = 2, ERR_HASH_FULL = 3, ERR_COUNT = 4 };
enum LeafKind { LEAF_NONE = 0, LEAF_VALUE = 1, LEAF_EVAL = 2 };

struct TreeDev {
    int32_t G;           // trees
    uint32_t R;          // slots per tree (reserve_space, src/node.rs:146)
    uint32_t H;          // hash entries per tree (power of two)
    // node store
    uint4* rec;          // [G*R] {link, prior bits, meta, child_base}
    uint64_t* ctr;       // [G*R] packed win counters (src/node.rs:17)
    ulonglong2* state;   // [G*R] canonical state {mine, theirs} (NodeMutableState.s)
    uint32_t* hash;      // [G*H] `seen`: open-addressing table of node indices (src/node.rs:135)
    uint32_t* len;       // [G] bump pointer (NodeStore.len, src/node.rs:134)
    // search state
    uint32_t* root;      // [G] root node of the current get_action_prob
    uint8_t* active;     // [G] tree takes part in the current search
    uint32_t* path;      // [G*PATH_CAP] node_path (src/async_mcts.rs:229)
    uint32_t* path_len;  // [G]
    uint32_t* leaf;      // [G] node the simulation stopped on
    uint32_t* leaf_kind; // [G] LeafKind
    float* leaf_val;     // [G] value when LEAF_VALUE
    int32_t* slot_of;    // [G] row of this tree's leaf in the eval batch (LEAF_EVAL)
    // diagnostics
    uint32_t* err;       // [ERR_COUNT]
    uint64_t* stat;      // [G*ST_COUNT]
    // eval log (replay parity): raw (pi, v) of every NNet::predict row, per tree, in order
    int32_t log_cap;
    uint32_t* log_len;      // [G]
    ulonglong2* log_state;  // [G*log_cap]
    float* log_pi;          // [G*log_cap*7]
    float* log_v;           // [G*log_cap]
};

// Compacted leaf batch handed to the net (src/async_mcts.rs:117-189 restated as lanes).
struct EvalBatch {
    int32_t cap;
    uint32_t* n;          // device: rows in this batch
    uint32_t* tree;       // [cap] row -> tree index
    ulonglong2* state;    // [cap] canonical state to featurise (to_features, connect_four_game.rs:219-237)
    float* pi;            // [cap*8] net output: pi[0..6], v repeated in slot 7
    float* v;             // [cap]
    // ---- leaf de-duplication (src == nullptr: off, every requested row is evaluated) -------------------------------
    // Thousands of games share their openings, and a row's (pi, v) depends on nothing but its state (BatchNorm is folded,
    // every row's K-sum has one order), so evaluating a state once per batch -- or once per call, through the engine's
    // evaluation cache -- is bit-exact.  The per-tree analogue in the reference is `seen` (src/node.rs:282-289).
    // k_dedup gives every requested row r a source src[r]: a row u of the UNIQUE batch (ustate/upi/uv, *un rows: what the
    // net really runs on), a slot of the batch's election table whose winner holds u, or an entry of the evaluation cache.
    uint32_t* src;        // [cap] SRC_* tagged
    uint32_t* un;         // device: unique rows of this batch
    ulonglong2* ustate;   // [cap]
    float* upi;           // [cap*8]
    float* uv;            // [cap]
    unsigned long long* tkey;   // [tmask+1] election table: (epoch << 49) | state key, stale epochs count as empty
    uint32_t* tuniq;      // [tmask+1] unique row of the slot's winner
    uint32_t tmask;
};
constexpr uint32_t SRC_TABLE = 0x80000000u, SRC_CACHE = 0x40000000u, SRC_INDEX = 0x3FFFFFFFu;

// Evaluation cache of the engine: state key (49 bits) | model tag (15 bits) -> (pi[7], v), 8-way buckets of one 64-byte key
// line.  Filled by the trees whose row was evaluated (in the backup kernels), read by k_dedup (a later kernel on the same
// stream), never evicted; entries of a model die with its tag (a new tag per weight upload).  key == nullptr: off.
struct EvalCache {
    unsigned long long* key;   // [(bmask+1)*8], 0 = empty
    float* pv;                 // [(bmask+1)*8][8]
    uint32_t bmask;            // buckets - 1
    uint32_t max_stones;       // only states with at most this many stones are inserted
    unsigned long long tag;    // model tag, already shifted to bits 49..63
    unsigned long long* stat;  // [DD_COUNT] requested / executed / cache hits / in-batch duplicates / inserts
};
enum DedupStat { DD_REQUESTED = 0, DD_EXECUTED, DD_CACHE_HITS, DD_BATCH_DUPS, DD_INSERTS, DD_COUNT };

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
    // per-episode outputs, indexed by gid
    int32_t n_games;
    ulonglong2* smp_state;   // [n_games*42]
    float* smp_pi;           // [n_games*42*7]
    int8_t* smp_player;      // [n_games*42]
    uint8_t* moves;          // [n_games*42]
    int32_t* g_len;          // [n_games]
    float* g_result;         // [n_games] r = get_game_ended(cur_player) at the end (src/coach.rs:144)
    int8_t* g_final_player;  // [n_games]
    // counters: [0] next episode to hand out, [1] episodes finished, [2] active slots
    uint32_t* counters;
};

struct SelfplayMoveParams {
    uint64_t seed;
    uint64_t first_game_id;
    int32_t temp_threshold;
    int32_t refill;          // hand finished slots the next episode
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
};

// ---- launchers (all asynchronous on `s`) --------------------------------------
void launch_reset_trees(const TreeDev& t, const uint8_t* flags /*[G] or nullptr = all*/, hipStream_t s,
                        const ulonglong2* roots = nullptr /*[G] root states, nullptr = initial board*/);
void launch_root_prepare(const TreeDev& t, const EvalBatch& eb, const ulonglong2* root_states, hipStream_t s);
void launch_select(const TreeDev& t, const EvalBatch& eb, SearchParams sp, hipStream_t s);
void launch_backup(const TreeDev& t, const EvalBatch& eb, const EvalCache& ec, int apply_only, hipStream_t s);
// backup of simulation i (batch eb_prev) + select of simulation i+1 (leaf appended to eb_next) in one launch
void launch_backup_select(const TreeDev& t, const EvalBatch& eb_prev, const EvalBatch& eb_next, const EvalCache& ec, SearchParams sp,
                          int apply_only, hipStream_t s);
// rows [0, *eb.n) -> src[] + the unique batch (eb.src != nullptr); epoch in [1, 32767], the caller clears tkey when it wraps
void launch_dedup(const EvalBatch& eb, const EvalCache& ec, uint32_t epoch, hipStream_t s);
void launch_root_policy(const TreeDev& t, float temp, uint64_t seed, uint64_t first_game_id, float* pi,
                        uint16_t* counts, float* q, hipStream_t s);
void launch_selfplay_move(const TreeDev& t, const GamesDev& gd, SelfplayMoveParams mp, hipStream_t s);
void launch_selfplay_sync_active(const TreeDev& t, const GamesDev& gd, hipStream_t s);
void launch_arena_sync(const TreeDev& t_new, const TreeDev& t_old, const ArenaDev& ad, hipStream_t s);
void launch_arena_move(const TreeDev& t, const ArenaDev& ad, uint64_t seed, hipStream_t s);
void launch_emit_samples(const GamesDev& gd, const int64_t* offsets, int symmetries, ulonglong2* out_states,
                         float* out_boards, float* out_pis, float* out_zs, hipStream_t s);

}  // namespace az
