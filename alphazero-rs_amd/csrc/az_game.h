// az_game.h -- the device-side `Game` seam (trait Game, src/game.rs:10-28).
//
// The reference's AsyncMcts<G> / NodeStore<G> are generic over a Game; here the tree kernels (az_tree.hip) are
// templates over a game POLICY type with the same nine responsibilities, restated for lanes:
//
//   trait Game (src/game.rs)                        policy member
//   ------------------------------------------------------------------------------------------------------------
//   get_init_board()                                State init()
//   get_feature_shape() / to_features()             FEATURES, feature(s, f)           f32 planes, NCHW order
//   get_next_state(1, a) + get_canonical_form()     State play(s, a)                  result is canonical again (B5)
//   get_valid_moves(1)                              uint32_t valid_mask(s)            bit a set <=> action a is legal
//   get_game_ended(1)                               uint32_t ended_code(s)            E_NONE / E_PLUS1 / E_MINUS1 / E_DRAW: e = -ended
//   get_symmetries()                                mirror(s), mirror_action(a)
//   eval_heuristic()                                (0 in the reference's one game, connect_four_game.rs:213-216; B10)
//   Eq + Hash (the `seen` map, src/node.rs:135)     hash(s), pack(s) / unpack(k)      pack is an 8-byte identity of the state
//   action count (get_action_size)                  ACTIONS, GROUP, MAX_PLIES
//
// A node keeps its state as the 8-byte `Packed` word (the record is 32 bytes: az_tree.h), so a policy must be able to
// pack a canonical state into 64 bits and back.  GROUP = lanes that serve one tree = slots of one child block: a power
// of two > ACTIONS (lane j evaluates child j, one lane is left for (pi, v) rows of ACTIONS + 1 floats).
//
// ConnectFour is the reference's one implementor (examples/connect_four_lib/connect_four_game.rs:81-238).
#pragma once
#include "az_common.h"

namespace az {

// ---- Connect Four on canonical bitboards (connect_four_game.rs:81-238) --------
// bit(col,row) = col*7 + row, row 0 = bottom; `mine` = side to move.
constexpr uint64_t C4_FULL = 0x3Full | (0x3Full << 7) | (0x3Full << 14) | (0x3Full << 21) | (0x3Full << 28) |
                             (0x3Full << 35) | (0x3Full << 42);
AZ_HD uint64_t c4_top(int c) { return 1ull << (c * 7 + 5); }
AZ_HD bool c4_has_four(uint64_t b) {
    uint64_t m;
    m = b & (b >> 1); if (m & (m >> 2)) return true;
    m = b & (b >> 7); if (m & (m >> 14)) return true;
    m = b & (b >> 6); if (m & (m >> 12)) return true;
    m = b & (b >> 8); if (m & (m >> 16)) return true;
    return false;
}
// valid-move bitmask (bit c set <=> heights[c] < 6), connect_four_game.rs:105-110
AZ_HD uint32_t c4_valid_mask(uint64_t mine, uint64_t theirs) {
    uint64_t mask = mine | theirs;
    uint32_t v = 0;
#pragma unroll
    for (int c = 0; c < 7; ++c) v |= (mask & c4_top(c)) ? 0u : (1u << c);
    return v;
}
// get_next_state(1, a) then get_canonical_form(next_player): connect_four_game.rs:90-103, :198-203 (B5)
AZ_HD void c4_play(uint64_t mine, uint64_t theirs, int a, uint64_t* nmine, uint64_t* ntheirs) {
    uint64_t mask = mine | theirs;
    uint64_t nb = (mask + (1ull << (a * 7))) & (0x3Full << (a * 7));
    *nmine = theirs;
    *ntheirs = mine | nb;
}
// ecode of a canonical state: e = -get_game_ended(1), connect_four_game.rs:112-196 (B6), src/node.rs:293-294
AZ_HD uint32_t c4_ecode(uint64_t mine, uint64_t theirs) {
    if (c4_has_four(mine)) return E_MINUS1;        // ended = +1 (unreachable in legal play)
    if (c4_has_four(theirs)) return E_PLUS1;       // ended = -1: the player who moved in has won
    if ((mine | theirs) == C4_FULL) return E_DRAW; // ended = DRAW_EPS
    return E_NONE;
}
AZ_HD uint64_t c4_mirror(uint64_t b) {
    uint64_t r = 0;
#pragma unroll
    for (int c = 0; c < 7; ++c) r |= ((b >> (c * 7)) & 0x7Full) << ((6 - c) * 7);
    return r;
}
// feature (plane, row-from-top, col) of a canonical state, connect_four_game.rs:219-237 (S8)
AZ_HD float c4_feature(uint64_t mine, uint64_t theirs, int plane, int r, int c) {
    uint64_t bit = 1ull << (c * 7 + (5 - r));
    return ((plane == 0 ? mine : theirs) & bit) ? 1.0f : 0.0f;
}
AZ_HD uint32_t c4_hash(uint64_t mine, uint64_t theirs) { return (uint32_t)mix64(mine ^ mix64(theirs)); }
// One-word identity of a canonical state (49 bits, never 0): mask + bottom row puts a single 1 above every column's stones,
// adding `mine` fills in the mover's stones below it (no carries).  c4_unkey inverts it column by column.
constexpr uint64_t C4_BOTTOM = 1ull | (1ull << 7) | (1ull << 14) | (1ull << 21) | (1ull << 28) | (1ull << 35) | (1ull << 42);
AZ_HD uint64_t c4_key(uint64_t mine, uint64_t theirs) { return mine + (mine | theirs) + C4_BOTTOM; }
AZ_HD void c4_unkey(uint64_t key, uint64_t* mine, uint64_t* theirs) {
    uint64_t m = 0, mask = 0;
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        const uint32_t col = (uint32_t)(key >> (c * 7)) & 0x7Fu;      // (1 << height) | mine's stones of the column
        const uint32_t top = 1u << (31 - __builtin_clz(col | 1u));
        m |= (uint64_t)(col ^ top) << (c * 7);
        mask |= (uint64_t)(top - 1u) << (c * 7);
    }
    *mine = m;
    *theirs = mask ^ m;
}

struct ConnectFour {
    static constexpr int ACTIONS = 7;          // connect_four_game.rs:14
    static constexpr int GROUP = 8;
    static constexpr int MAX_PLIES = 42;
    static constexpr int FEATURES = 84;        // [2,6,7], connect_four_game.rs:86-88
    using State = ulonglong2;                  // {mine, theirs}
    using Packed = unsigned long long;
    AZ_HD static State init() { return make_ulonglong2(0ull, 0ull); }
    AZ_HD static State play(State s, int a) {
        uint64_t m, t;
        c4_play(s.x, s.y, a, &m, &t);
        return make_ulonglong2(m, t);
    }
    AZ_HD static uint32_t valid_mask(State s) { return c4_valid_mask(s.x, s.y); }
    AZ_HD static uint32_t ended_code(State s) { return c4_ecode(s.x, s.y); }
    AZ_HD static uint32_t hash(State s) { return c4_hash(s.x, s.y); }
    AZ_HD static Packed pack(State s) { return c4_key(s.x, s.y); }
    AZ_HD static State unpack(Packed k) {
        uint64_t m, t;
        c4_unkey(k, &m, &t);
        return make_ulonglong2(m, t);
    }
    AZ_HD static uint32_t stones(State s) {
#if defined(__HIP_DEVICE_COMPILE__)
        return (uint32_t)__popcll(s.x | s.y);
#else
        return (uint32_t)__builtin_popcountll(s.x | s.y);
#endif
    }
    AZ_HD static State mirror(State s) { return make_ulonglong2(c4_mirror(s.x), c4_mirror(s.y)); }   // get_symmetries, :205-211
    AZ_HD static int mirror_action(int a) { return ACTIONS - 1 - a; }
    AZ_HD static float feature(State s, int f) { return c4_feature(s.x, s.y, f / 42, (f % 42) / 7, f % 7); }
};

// The seam's SECOND instantiation: Connect Four's board, moves, features and symmetries with three in a row winning.  The
// reference has no second Game; this one exists so that the Game policy seam is exercised end to end (az_config.game = 1,
// oracle twin CBits<3>): the tree kernels below it are the same templates, nothing in them names a rule.
AZ_HD bool c4_has_three(uint64_t b) {
    return ((b & (b >> 1) & (b >> 2)) | (b & (b >> 7) & (b >> 14)) | (b & (b >> 6) & (b >> 12)) | (b & (b >> 8) & (b >> 16))) != 0ull;
}
struct ConnectThree : ConnectFour {
    AZ_HD static uint32_t ended_code(State s) {
        if (c4_has_three(s.x)) return E_MINUS1;
        if (c4_has_three(s.y)) return E_PLUS1;
        if ((s.x | s.y) == C4_FULL) return E_DRAW;
        return E_NONE;
    }
};

// test-fixture net (exact in f32) on a 16-byte state; oracle twin: hashnet_eval in oracle/az_oracle_games.hpp
AZ_HD void hashnet_eval(uint64_t mine, uint64_t theirs, uint64_t salt, float* pi, float* v) {
    uint64_t h = mix64(mine ^ mix64(theirs ^ mix64(salt)));
#pragma unroll
    for (int a = 0; a < 7; ++a)
        pi[a] = (float)(uint32_t)((mix64(h + (uint64_t)a) >> 40) + 1) * (1.0f / 16777216.0f);
    *v = (float)(uint32_t)(mix64(h + 7) >> 40) * (1.0f / 8388608.0f) - 1.0f;
}

}  // namespace az
