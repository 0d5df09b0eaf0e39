// az_oracle_games.hpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
//
// Games for the oracle: Connect Four in two independent representations
// (7x6 bitboards, and the reference's i8[6][7] + heights arrays) that are
// cross-checked against each other, the reference's DummyGame test fixture,
// the episode driver and the arena.  See az_oracle.hpp for the parity status.
#pragma once
#include "az_oracle.hpp"

namespace azo {

constexpr int C4_H = 6;        // connect_four_game.rs:13
constexpr int C4_W = 7;        // :14
constexpr int C4_WIN = 4;      // :15
constexpr float DRAW_EPS = 1e-4f; // :16

// ---------------------------------------------------------------------------
// Bitboard Connect Four.  bit(col,row) = col*7 + row, row 0 = BOTTOM, the 7th
// bit of every column is a permanently clear sentinel.  `p1` holds player +1's
// stones, `m1` player -1's.  In canonical form (side to move == +1) this is the
// (mine, theirs) pair the HIP engine stores per node.
// WIN = 4 is Connect Four (the reference's one Game); WIN = 3 ("Connect Three": same board, moves and features, three in a
// row wins) is the twin of the engine's second Game policy, kept to exercise the Game seam end to end.
template <int WIN>
struct CBits {
    uint64_t p1 = 0, m1 = 0;

    static constexpr uint64_t col_mask(int c) { return 0x3Full << (c * 7); }
    static constexpr uint64_t top_bit(int c) { return 1ull << (c * 7 + 5); }
    static constexpr uint64_t bottom_bit(int c) { return 1ull << (c * 7); }
    static constexpr uint64_t FULL = 0x3Full | (0x3Full << 7) | (0x3Full << 14) | (0x3Full << 21) |
                                     (0x3Full << 28) | (0x3Full << 35) | (0x3Full << 42);
    static bool has_four(uint64_t b) {
        // all 69 windows (B6 repair): vertical 1, horizontal 7, diagonals 6 and 8
        uint64_t m;
        if (WIN == 3) {
            for (int sft : {1, 7, 6, 8}) if (b & (b >> sft) & (b >> (2 * sft))) return true;
            return false;
        }
        m = b & (b >> 1); if (m & (m >> 2)) return true;
        m = b & (b >> 7); if (m & (m >> 14)) return true;
        m = b & (b >> 6); if (m & (m >> 12)) return true;
        m = b & (b >> 8); if (m & (m >> 16)) return true;
        return false;
    }

    struct Hasher {
        size_t operator()(const CBits& g) const { return (size_t)mix64(g.p1 ^ mix64(g.m1)); }
    };
    bool operator==(const CBits& o) const { return p1 == o.p1 && m1 == o.m1; }

    static CBits get_init_board() { return CBits{}; }                 // connect_four_game.rs:82-84
    static std::vector<size_t> get_feature_shape() { return {2, C4_H, C4_W}; } // :86-88 (S8: NCHW)
    // :90-103
    std::pair<CBits, int8_t> get_next_state(int8_t player, uint8_t action) const {
        uint64_t mask = p1 | m1;
        uint64_t nb = (mask + bottom_bit(action)) & col_mask(action);
        assert(nb != 0);
        CBits n = *this;
        if (player == 1) n.p1 |= nb; else n.m1 |= nb;
        return {n, (int8_t)-player};
    }
    // :105-110
    std::vector<uint8_t> get_valid_moves(int8_t) const {
        uint64_t mask = p1 | m1;
        std::vector<uint8_t> v(C4_W);
        for (int c = 0; c < C4_W; ++c) v[c] = (mask & top_bit(c)) ? 0 : 1;
        return v;
    }
    // :112-196 with B6 repaired
    float get_game_ended(int8_t player) const {
        if (has_four(p1)) return player == 1 ? 1.0f : -1.0f;
        if (has_four(m1)) return player == -1 ? 1.0f : -1.0f;
        if ((p1 | m1) == FULL) return DRAW_EPS;
        return 0.0f;
    }
    // :198-203 with B5 repaired: board from the side-to-move's point of view
    CBits get_canonical_form(int8_t player) const {
        return player == 1 ? *this : CBits{m1, p1};
    }
    static uint64_t mirror(uint64_t b) {
        uint64_t r = 0;
        for (int c = 0; c < C4_W; ++c) r |= ((b >> (c * 7)) & 0x7Full) << ((C4_W - 1 - c) * 7);
        return r;
    }
    CBits flip() const { return CBits{mirror(p1), mirror(m1)}; }      // :65-78
    // :205-211
    std::vector<std::pair<CBits, std::vector<float>>> get_symmetries(const std::vector<float>& pi) const {
        std::vector<float> rp(pi.rbegin(), pi.rend());
        return {{*this, pi}, {flip(), rp}};
    }
    float eval_heuristic() const { return 0.0f; }                       // :213-216
    // :219-237 with S8 (A9): [2,6,7], plane 0 = +1 stones, plane 1 = -1 stones, row 0 = top
    std::vector<float> to_features() const {
        std::vector<float> f(2 * C4_H * C4_W, 0.0f);
        for (int r = 0; r < C4_H; ++r)
            for (int c = 0; c < C4_W; ++c) {
                uint64_t bit = 1ull << (c * 7 + (C4_H - 1 - r));
                if (p1 & bit) f[0 * 42 + r * 7 + c] = 1.0f;
                if (m1 & bit) f[1 * 42 + r * 7 + c] = 1.0f;
            }
        return f;
    }
};
using C4Bits = CBits<4>;
using C3Bits = CBits<3>;

// ---------------------------------------------------------------------------
// Array Connect Four: the reference's own data layout (i8[6][7], row 0 = top,
// heights[7]) -- connect_four_game.rs:18-23 -- with B5/B6 repaired; B6 can be
// switched back to the literal loops to show which windows the reference misses.
struct C4Array {
    int8_t s[C4_H][C4_W];
    uint8_t heights[C4_W];
    bool literal_windows = false;

    C4Array() { std::memset(s, 0, sizeof(s)); std::memset(heights, 0, sizeof(heights)); }
    struct Hasher {
        size_t operator()(const C4Array& g) const {
            uint64_t h = 0;
            for (int r = 0; r < C4_H; ++r)
                for (int c = 0; c < C4_W; ++c) h = mix64(h ^ (uint64_t)(uint8_t)g.s[r][c]);
            return (size_t)h;
        }
    };
    bool operator==(const C4Array& o) const { return std::memcmp(s, o.s, sizeof(s)) == 0; } // :48-52
    static C4Array get_init_board() { return C4Array(); }
    static std::vector<size_t> get_feature_shape() { return {2, C4_H, C4_W}; }
    std::pair<C4Array, int8_t> get_next_state(int8_t player, uint8_t action) const { // :90-103
        C4Array n = *this;
        assert(n.heights[action] < C4_H);
        n.heights[action] += 1;
        n.s[C4_H - n.heights[action]][action] = player;
        return {n, (int8_t)-player};
    }
    std::vector<uint8_t> get_valid_moves(int8_t) const {              // :105-110
        std::vector<uint8_t> v(C4_W);
        for (int c = 0; c < C4_W; ++c) v[c] = heights[c] < C4_H ? 1 : 0;
        return v;
    }
    float get_game_ended(int8_t player) const {                       // :112-196
        const int hend = literal_windows ? C4_W - C4_WIN : C4_W - C4_WIN + 1;  // :114 `0..W-L`
        const int vend = literal_windows ? C4_H - C4_WIN : C4_H - C4_WIN + 1;  // :129 `0..H-L`
        for (int row = 0; row < C4_H; ++row)
            for (int col = 0; col < hend; ++col) {
                int8_t x = s[row][col];
                if (x != 0 && x == s[row][col + 1] && x == s[row][col + 2] && x == s[row][col + 3])
                    return player == x ? 1.0f : -1.0f;
            }
        for (int row = 0; row < vend; ++row)
            for (int col = 0; col < C4_W; ++col) {
                int8_t x = s[row][col];
                if (x != 0 && x == s[row + 1][col] && x == s[row + 2][col] && x == s[row + 3][col])
                    return player == x ? 1.0f : -1.0f;
            }
        for (int row = 0; row <= C4_H - C4_WIN; ++row)                // :150-151
            for (int col = 0; col <= C4_W - C4_WIN; ++col) {
                int8_t x = s[row][col];
                if (x != 0 && x == s[row + 1][col + 1] && x == s[row + 2][col + 2] && x == s[row + 3][col + 3])
                    return player == x ? 1.0f : -1.0f;
            }
        for (int row = 0; row <= C4_H - C4_WIN; ++row)                // :171-172
            for (int col = C4_WIN - 1; col < C4_W; ++col) {
                int8_t x = s[row][col];
                if (x != 0 && x == s[row + 1][col - 1] && x == s[row + 2][col - 2] && x == s[row + 3][col - 3])
                    return player == x ? 1.0f : -1.0f;
            }
        for (int c = 0; c < C4_W; ++c) if (heights[c] < C4_H) return 0.0f;
        return DRAW_EPS;                                              // :191-195
    }
    C4Array get_canonical_form(int8_t player) const {                 // :198-203, B5 repaired
        C4Array n = *this;
        for (int r = 0; r < C4_H; ++r) for (int c = 0; c < C4_W; ++c) n.s[r][c] = (int8_t)(s[r][c] * player);
        return n;
    }
    C4Array flip() const {                                            // :65-78
        C4Array n = *this;
        for (int r = 0; r < C4_H; ++r) for (int c = 0; c < C4_W; ++c) n.s[r][c] = s[r][C4_W - 1 - c];
        for (int c = 0; c < C4_W; ++c) n.heights[c] = heights[C4_W - 1 - c];
        return n;
    }
    std::vector<std::pair<C4Array, std::vector<float>>> get_symmetries(const std::vector<float>& pi) const {
        std::vector<float> rp(pi.rbegin(), pi.rend());
        return {{*this, pi}, {flip(), rp}};
    }
    float eval_heuristic() const { return 0.0f; }
    std::vector<float> to_features() const {                          // :219-237, S8
        std::vector<float> f(2 * C4_H * C4_W, 0.0f);
        for (int r = 0; r < C4_H; ++r)
            for (int c = 0; c < C4_W; ++c) {
                if (s[r][c] == 1) f[0 * 42 + r * 7 + c] = 1.0f;
                if (s[r][c] == -1) f[1 * 42 + r * 7 + c] = 1.0f;
            }
        return f;
    }
    C4Bits to_bits() const {
        C4Bits b;
        for (int r = 0; r < C4_H; ++r)
            for (int c = 0; c < C4_W; ++c) {
                uint64_t bit = 1ull << (c * 7 + (C4_H - 1 - r));
                if (s[r][c] == 1) b.p1 |= bit;
                if (s[r][c] == -1) b.m1 |= bit;
            }
        return b;
    }
};

// ---------------------------------------------------------------------------
// DummyGame: src/node/tests/dummy_game.rs:11-84 (fixture of the node.rs tests)
struct DummyGame {
    uint8_t _s = 0;
    DummyGame() = default;
    explicit DummyGame(uint8_t v) : _s(v) {}
    struct Hasher { size_t operator()(const DummyGame& g) const { return g._s; } };
    bool operator==(const DummyGame& o) const { return _s == o._s; }
    static DummyGame get_init_board() { return DummyGame(0); }
    static std::vector<size_t> get_feature_shape() { return {1}; }
    std::pair<DummyGame, int8_t> get_next_state(int8_t player, uint8_t) const {
        return {DummyGame((uint8_t)(_s + 1)), (int8_t)(1 - player)};
    }
    std::vector<uint8_t> get_valid_moves(int8_t) const { return {0}; }
    float get_game_ended(int8_t) const { return 0.0f; }
    DummyGame get_canonical_form(int8_t) const { return DummyGame(0); }
    std::vector<std::pair<DummyGame, std::vector<float>>> get_symmetries(const std::vector<float>& pi) const {
        return {{DummyGame(_s), pi}};
    }
    float eval_heuristic() const { return 0.0f; }
    std::vector<float> to_features() const { return {(float)_s}; }
};

// ---------------------------------------------------------------------------
// Nets of the oracle.
// S9 (A10): DumbConnectFourNnet, examples/connect_four.rs:12-43: pi = 1/width, v = +1.
struct StubNet : NNet {
    void predict(const float*, int B, int, float* pi, float* v) override {
        for (int b = 0; b < B; ++b) {
            for (int a = 0; a < C4_W; ++a) pi[b * C4_W + a] = 1.0f / (float)C4_W;
            v[b] = 1.0f;
        }
    }
};
// Deterministic pseudo-random net (no reference counterpart): exercises the
// tree with non-uniform priors/values using only exactly representable floats,
// so the HIP engine reproduces it bit for bit.  Keyed on the feature planes.
inline void hashnet_eval(uint64_t mine, uint64_t theirs, uint64_t salt, float* pi, float* v) {
    uint64_t h = mix64(mine ^ mix64(theirs ^ mix64(salt)));
    for (int a = 0; a < C4_W; ++a)
        pi[a] = (float)(uint32_t)((mix64(h + (uint64_t)a) >> 40) + 1) * (1.0f / 16777216.0f);
    *v = (float)(uint32_t)(mix64(h + 7) >> 40) * (1.0f / 8388608.0f) - 1.0f;
}
inline void features_to_bits(const float* f, uint64_t* mine, uint64_t* theirs) {
    uint64_t a = 0, b = 0;
    for (int r = 0; r < C4_H; ++r)
        for (int c = 0; c < C4_W; ++c) {
            uint64_t bit = 1ull << (c * 7 + (C4_H - 1 - r));
            if (f[r * 7 + c] != 0.0f) a |= bit;
            if (f[42 + r * 7 + c] != 0.0f) b |= bit;
        }
    *mine = a; *theirs = b;
}
struct HashNet : NNet {
    uint64_t salt = 0;
    bool per_model = true;  // model_id is mixed into the salt (two arena nets differ)
    void predict(const float* boards, int B, int model_id, float* pi, float* v) override {
        for (int b = 0; b < B; ++b) {
            uint64_t m, t;
            features_to_bits(boards + (size_t)b * 84, &m, &t);
            hashnet_eval(m, t, salt + (per_model ? (uint64_t)model_id * 0x51ED27ull : 0), pi + b * C4_W, v + b);
        }
    }
};
// Replays (pi, v) records produced elsewhere (the HIP engine's bf16 net) in the
// order the search asks for them, checking the state each record was made for.
struct ReplayNet : NNet {
    const uint64_t* states = nullptr;  // [n,2] (mine, theirs), may be null
    const float* pis = nullptr;        // [n,7]
    const float* vs = nullptr;         // [n]
    size_t n = 0, pos = 0;
    bool mismatch = false;
    long first_bad = -1;      // record index of the first request that did not match its record (diagnostics)
    uint64_t bad_req[2] = {0, 0};   // ... and the state that was requested there
    void predict(const float* boards, int B, int, float* pi, float* v) override {
        for (int b = 0; b < B; ++b) {
            if (pos >= n) { mismatch = true; if (first_bad < 0) first_bad = (long)pos; for (int a = 0; a < C4_W; ++a) pi[b * 7 + a] = 1.0f / 7.0f; v[b] = 0; continue; }
            if (states) {
                uint64_t m, t;
                features_to_bits(boards + (size_t)b * 84, &m, &t);
                if (m != states[2 * pos] || t != states[2 * pos + 1]) { mismatch = true; if (first_bad < 0) { first_bad = (long)pos; bad_req[0] = m; bad_req[1] = t; } }
            }
            for (int a = 0; a < C4_W; ++a) pi[b * 7 + a] = pis[pos * 7 + a];
            v[b] = vs[pos];
            ++pos;
        }
    }
};

// ---------------------------------------------------------------------------
// Coach::execute_episode, src/coach.rs:104-157 (C15, B4, B7).
template <class G>
std::vector<TrainingSample> execute_episode(AsyncMcts<G>& mcts, size_t temp_threshold, uint64_t seed,
                                            uint64_t game_id, std::vector<uint8_t>* moves_out = nullptr) {
    struct Ex { std::vector<float> f; int8_t player; std::vector<float> pi; };
    std::vector<Ex> train_examples;
    G board = G::get_init_board();
    int8_t cur_player = 1;
    size_t episode_step = 0;
    for (;;) {
        episode_step += 1;                                              // :119
        G canonical = board.get_canonical_form(cur_player);             // :120
        float temp = episode_step < temp_threshold ? 1.0f : 0.0f;       // :122-126
        uint64_t ply = episode_step - 1;
        std::vector<float> pi = mcts.get_action_prob(canonical, temp, seed, game_id, ply); // :128
        for (auto& bp : canonical.get_symmetries(pi))                   // :130-135
            train_examples.push_back({bp.first.to_features(), cur_player, bp.second});
        uint64_t r64 = rng_draw(seed, game_id, ply, RNG_MOVE);
        uint8_t action = (uint8_t)rng_choose_weighted(r64, pi.data(), (int)pi.size()); // :137-138
        if (moves_out) moves_out->push_back(action);
        auto nx = board.get_next_state(cur_player, action);             // :140-142
        board = nx.first;
        cur_player = nx.second;
        float r = board.get_game_ended(cur_player);                     // :144
        if (r != 0.0f) {
            std::vector<TrainingSample> out;
            for (auto& ex : train_examples) {
                float z;
                if (mcts.quirks.b4_literal_z) z = ex.player == cur_player ? 1.0f : -1.0f; // :152
                else z = r * (ex.player == cur_player ? 1.0f : -1.0f);   // B4
                out.push_back({ex.f, ex.pi, z});
            }
            return out;
        }
    }
}

// arena::play_games, src/arena.rs:62-99 (C16) with per-game trees (B8 repair):
// make_player(model_slot, game_index) builds a fresh searcher for every game.
// Seating order: Heap's permutations of [new, old] = (new, old) then (old, new).
template <class G>
GameResultCounts play_games(size_t num,
                            const std::function<std::function<uint8_t(const G&)>(int, size_t)>& make_player,
                            const std::optional<G>& board) {
    GameResultCounts all;
    size_t game_index = 0;
    for (int ordering = 0; ordering < 2; ++ordering) {
        int first = ordering == 0 ? 0 : 1;                              // player_ordering[0].0
        int win_cond = first == 0 ? 1 : -1;                             // :80
        int lose_cond = first == 0 ? -1 : 1;                            // :81
        for (size_t i = 0; i < num / 2; ++i, ++game_index) {            // :83
            std::function<uint8_t(const G&)> acts[2] = {make_player(first, game_index),
                                                         make_player(1 - first, game_index)};
            int8_t res = play_game<G>(acts, board);
            if (res == win_cond) all.win++;
            else if (res == lose_cond) all.loss++;
            else all.draw++;
        }
    }
    return all;
}

}  // namespace azo
