// az_host.hpp -- C++ host-side mirror of the reference's plugin interface above the C ABI.
//
// The reference is Rust; no Rust toolchain exists in this build environment, so the host side of the
// drop-in is written in C++ with the reference's names, argument meaning and error behaviour (a Rust
// `panic!` becomes a thrown az_host::Panic):
//   Game trait            src/game.rs:10-28           -> ConnectFourGame (connect_four_game.rs:81-238)
//   NNet trait            src/nnet.rs:35-45           -> NNet, Mi355xNNet (az_net_* entry points)
//   AsyncMcts<G>          src/async_mcts.rs:14-115    -> AsyncMcts (az_tree_* entry points, one tree)
//   arena::play_game(s)   src/arena.rs:7-99           -> play_game, play_games (closures, Heap's order)
//   Coach::execute_episode src/coach.rs:104-157       -> execute_episode
//   Coach::setup / learn  src/coach.rs:38-103, :169-396 -> Coach (az_selfplay, az_net_train, az_arena; one call each)
// Header-only; links against libaz_engine.so.  Nothing here touches oracle/.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <filesystem>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "az_engine.h"

namespace az_host {

struct Panic : std::runtime_error { using std::runtime_error::runtime_error; };

using Policy = std::vector<float>;          // nnet.rs:17
using BoardFeatures = std::vector<float>;   // nnet.rs:9, [2,6,7] row-major
struct TrainingSample { BoardFeatures board; Policy pi; float v; };   // nnet.rs:22-27

// ---- Game: Connect Four on absolute bitboards (player +1 / -1) ------------------------------------------
class ConnectFourGame {
  public:
    uint64_t plus = 0, minus = 0;           // stones of player +1 / -1; bit(col,row) = col*7+row, row 0 = bottom
    static constexpr int H = 6, W = 7;      // connect_four_game.rs:13-14
    static constexpr float DRAW_EPS = 1e-4f;

    static ConnectFourGame get_init_board() { return {}; }
    static std::vector<size_t> get_feature_shape() { return {2, 6, 7}; }
    std::pair<ConnectFourGame, int8_t> get_next_state(int8_t player, uint8_t action) const {
        const uint64_t mask = plus | minus;
        const uint64_t nb = (mask + (1ull << (action * 7))) & (0x3Full << (action * 7));
        if (!nb) throw Panic("C4 move into a full column");           // debug_assert, connect_four_game.rs:98
        ConnectFourGame n = *this;
        (player == 1 ? n.plus : n.minus) |= nb;
        return {n, (int8_t)-player};
    }
    std::array<uint8_t, 7> get_valid_moves(int8_t) const {
        std::array<uint8_t, 7> v{};
        for (int c = 0; c < W; ++c) v[c] = ((plus | minus) >> (c * 7 + 5)) & 1 ? 0 : 1;
        return v;
    }
    float get_game_ended(int8_t player) const {
        if (four(plus)) return player == 1 ? 1.f : -1.f;
        if (four(minus)) return player == -1 ? 1.f : -1.f;
        for (int c = 0; c < W; ++c) if (!(((plus | minus) >> (c * 7 + 5)) & 1)) return 0.f;
        return DRAW_EPS;
    }
    ConnectFourGame get_canonical_form(int8_t player) const {         // side to move becomes +1 (B5)
        ConnectFourGame n;
        n.plus = player == 1 ? plus : minus;
        n.minus = player == 1 ? minus : plus;
        return n;
    }
    std::vector<std::pair<ConnectFourGame, Policy>> get_symmetries(const Policy& pi) const {
        ConnectFourGame f;
        for (int c = 0; c < W; ++c) {
            f.plus |= ((plus >> (c * 7)) & 0x7F) << ((6 - c) * 7);
            f.minus |= ((minus >> (c * 7)) & 0x7F) << ((6 - c) * 7);
        }
        return {{*this, pi}, {f, Policy(pi.rbegin(), pi.rend())}};
    }
    float eval_heuristic() const { return 0.f; }
    BoardFeatures to_features() const {                                // [2,6,7], row 0 = top (S8)
        BoardFeatures f(84, 0.f);
        for (int r = 0; r < H; ++r)
            for (int c = 0; c < W; ++c) {
                const uint64_t bit = 1ull << (c * 7 + (5 - r));
                if (plus & bit) f[r * 7 + c] = 1.f;
                if (minus & bit) f[42 + r * 7 + c] = 1.f;
            }
        return f;
    }
    bool operator==(const ConnectFourGame& o) const { return plus == o.plus && minus == o.minus; }

  private:
    static bool four(uint64_t b) {
        for (int d : {1, 7, 6, 8}) { uint64_t m = b & (b >> d); if (m & (m >> (2 * d))) return true; }
        return false;
    }
};

// ---- engine handle + NNet ----------------------------------------------------------------------------------
class Engine {
  public:
    explicit Engine(int device = 0, int max_batch = 8192, int channels = 512) {
        az_config cfg{device, max_batch, channels, 0, 0};
        if (az_create(&cfg, &e_) != AZ_OK) throw Panic("az_create failed");
    }
    ~Engine() { az_destroy(e_); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;
    az_engine* raw() const { return e_; }
    void check(int rc) const { if (rc != AZ_OK) throw Panic(std::string(az_last_error(e_))); }

  private:
    az_engine* e_ = nullptr;
};

class NNet {                                                            // src/nnet.rs:35-45
  public:
    virtual ~NNet() = default;
    virtual void train(const float* boards, const float* pis, const float* vs, int64_t n, size_t prev_id, size_t id) = 0;
    virtual void predict(const float* boards, int batch, size_t model_id, float* pi, float* v) const = 0;
};
class Mi355xNNet : public NNet {
  public:
    explicit Mi355xNNet(Engine& e) : e_(e) {}
    void train(const float* b, const float* p, const float* v, int64_t n, size_t prev_id, size_t id) override {
        e_.check(az_net_train(e_.raw(), (int)prev_id, (int)id, b, p, v, n));
    }
    void predict(const float* boards, int batch, size_t model_id, float* pi, float* v) const override {
        e_.check(az_net_predict(e_.raw(), (int)model_id, boards, batch, pi, v));
    }

  private:
    Engine& e_;
};

// ---- AsyncMcts: one tree behind az_tree_* ----------------------------------------------------------------------
class AsyncMcts {
  public:
    // AsyncMcts::default(reserve_space, num_sims, num_threads, max_depth, model_id, cpuct, ..), src/async_mcts.rs:27-48
    // num_threads > 1: several simulations in flight per tree, the engine's deterministic lock-step schedule (az_engine.h)
    static AsyncMcts default_(Engine& e, size_t reserve_space, size_t num_sims, size_t num_threads, size_t max_depth,
                              size_t model_id, int32_t cpuct) {
        if (num_threads == 0 || num_sims % num_threads != 0)
            throw Panic("assertion failed: self.num_sims % self.num_threads == 0");   // src/async_mcts.rs:192
        return AsyncMcts(e, reserve_space, num_sims, num_threads, max_depth, model_id, cpuct);
    }
    AsyncMcts(AsyncMcts&& o) noexcept : e_(o.e_), t_(o.t_) { o.t_ = nullptr; }
    ~AsyncMcts() { if (t_) az_tree_destroy(t_); }
    // get_action_prob(&self, s, temp, episode_id, rng): `s` canonical; rng = (seed, episode_id) stream (B7)
    Policy get_action_prob(const ConnectFourGame& s, float temp, size_t episode_id, uint64_t seed,
                           std::array<uint16_t, 7>* counts = nullptr, std::array<float, 7>* q = nullptr) const {
        const uint64_t st[2] = {s.plus, s.minus};
        Policy pi(7);
        e_.check(az_tree_get_action_prob(t_, st, temp, seed, (uint64_t)episode_id, pi.data(),
                                         counts ? counts->data() : nullptr, q ? q->data() : nullptr));
        return pi;
    }

  private:
    AsyncMcts(Engine& e, size_t reserve, size_t sims, size_t threads, size_t max_depth, size_t model_id, int32_t cpuct) : e_(e) {
        e_.check(az_tree_create(e.raw(), 1, reserve, (int)sims, (int)threads, (int)max_depth, (int)model_id, cpuct, &t_));
    }
    Engine& e_;
    az_tree* t_ = nullptr;
};

// ---- arena (src/arena.rs:7-99) ------------------------------------------------------------------------------------
using PlayerAction = std::function<uint8_t(const ConnectFourGame&)>;

inline int8_t play_game(const std::array<const PlayerAction*, 2>& player_actions, const ConnectFourGame* board0) {
    int8_t cur_player = 1;
    ConnectFourGame board = board0 ? *board0 : ConnectFourGame::get_init_board();
    while (board.get_game_ended(cur_player) == 0.f) {                          // :18
        const ConnectFourGame canonical = board.get_canonical_form(cur_player); // :25
        const uint8_t action = (*player_actions[cur_player == 1 ? 0 : 1])(canonical);
        if (canonical.get_valid_moves(1)[action] == 0) throw Panic("Action is not valid!");   // :31-35
        auto nx = board.get_next_state(cur_player, action);
        board = nx.first;
        cur_player = nx.second;
    }
    return (int8_t)(cur_player * (int8_t)std::lround(board.get_game_ended(cur_player)));      // :51
}

struct GameResultCounter { size_t win = 0, loss = 0, draw = 0; };             // Counter<GameResult>, :54-59

inline GameResultCounter play_games(size_t num, const std::array<const PlayerAction*, 2>& player_actions,
                                    const ConnectFourGame* board) {
    GameResultCounter all;
    for (int ordering = 0; ordering < 2; ++ordering) {                         // Heap's permutations of 2: (0,1), (1,0)
        const std::array<const PlayerAction*, 2> seated = {player_actions[ordering], player_actions[1 - ordering]};
        const int win_cond = ordering == 0 ? 1 : -1, lose_cond = -win_cond;    // :80-81
        for (size_t i = 0; i < num / 2; ++i) {                                 // :83
            const int8_t r = play_game(seated, board);
            if (r == win_cond) ++all.win; else if (r == lose_cond) ++all.loss; else ++all.draw;
        }
    }
    return all;
}

// ---- Coach::execute_episode (src/coach.rs:104-157) with the engine's RNG stream --------------------------------------
inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint8_t choose_weighted(uint64_t seed, uint64_t episode_id, uint64_t ply, const Policy& pi) {
    const uint64_t r = mix64(mix64(mix64(mix64(seed) ^ episode_id) ^ ply) ^ 2ull);
    float total = 0.f;
    for (float p : pi) total = total + p;
    const float target = (float)(uint32_t)(r >> 40) * (1.0f / 16777216.0f) * total;
    float acc = 0.f;
    int last = 0;
    for (int a = 0; a < (int)pi.size(); ++a)
        if (pi[a] > 0.f) { acc = acc + pi[a]; last = a; if (target < acc) return (uint8_t)a; }
    return (uint8_t)last;
}

inline std::vector<TrainingSample> execute_episode(const AsyncMcts& mcts, size_t temp_threshold, size_t episode_id,
                                                   uint64_t seed, std::vector<uint8_t>* moves = nullptr) {
    struct Ex { BoardFeatures f; int8_t player; Policy pi; };
    std::vector<Ex> train_examples;
    ConnectFourGame board = ConnectFourGame::get_init_board();
    int8_t cur_player = 1;
    size_t episode_step = 0;
    for (;;) {
        ++episode_step;
        const ConnectFourGame canonical = board.get_canonical_form(cur_player);
        const float temp = episode_step < temp_threshold ? 1.f : 0.f;             // :122-126
        const Policy pi = mcts.get_action_prob(canonical, temp, episode_id, seed);   // :128
        for (auto& bp : canonical.get_symmetries(pi)) train_examples.push_back({bp.first.to_features(), cur_player, bp.second});
        const uint8_t action = choose_weighted(seed, episode_id, episode_step - 1, pi);   // :137-138
        if (moves) moves->push_back(action);
        auto nx = board.get_next_state(cur_player, action);
        board = nx.first;
        cur_player = nx.second;
        const float r = board.get_game_ended(cur_player);                            // :144
        if (r != 0.f) {
            std::vector<TrainingSample> out;
            for (auto& ex : train_examples) out.push_back({ex.f, ex.pi, r * (ex.player == cur_player ? 1.f : -1.f)});   // B4
            return out;
        }
    }
}

// ---- Coach (src/coach.rs:17-396) ----------------------------------------------------------------------------------
// The iteration loop around the engine: self-play episodes -> replay window -> <iter>.examples -> shuffle ->
// NNet::train(samples, model_id, model_id + 1) -> arena of new vs old -> accept iff nwins + pwins > 0 and
// nwins / (nwins + pwins) >= update_threshold.  Self-play, training and the arena are ONE engine call each.
// The Python host (alphazero-rs_amd/coach.py) runs the same sequence with the same seeds and writes the same files.
//
// <dir>/<iter>.examples ("AZEX0001", the build's own format; the reference's is bincode, src/coach.rs:159-167, A14):
//   char magic[8]; int64 H; int64 lens[H]; f32 boards[N][2][6][7]; f32 pis[N][7]; f32 vs[N]   (N = sum lens; the whole
//   `history` deque, oldest entry first).   <dir>/<model_id>.aznet = az_net_save.
struct HistoryEntry { std::vector<float> boards, pis, vs; size_t len() const { return vs.size(); } };

// Fisher-Yates with the build's counter RNG (the reference shuffles with SmallRng, src/coach.rs:296-297; B7):
// for i = n-1 .. 1: j = ((draw(seed, iteration, i, 5) >> 32) * (i + 1)) >> 32; swap(perm[i], perm[j])
inline std::vector<int64_t> shuffle_permutation(size_t n, uint64_t seed, uint64_t iteration) {
    std::vector<int64_t> perm(n);
    for (size_t i = 0; i < n; ++i) perm[i] = (int64_t)i;
    for (size_t i = n; i-- > 1;) {
        const uint64_t r = mix64(mix64(mix64(mix64(seed) ^ iteration) ^ (uint64_t)i) ^ 5ull);
        const size_t j = (size_t)(((r >> 32) * (uint64_t)(i + 1)) >> 32);
        std::swap(perm[i], perm[j]);
    }
    return perm;
}

class Coach {
  public:
    struct Report { size_t iteration, samples, nwins, pwins, draws, model_id; bool accepted; std::vector<float> losses; };

    // Coach::setup(checkpoint_directory, + the reference's 14 numeric parameters), src/coach.rs:38-103
    static Coach setup(Engine& e, const std::string& checkpoint_directory, size_t mcts_reserve_size, float update_threshold,
                       size_t temp_threshold, size_t max_history_length, size_t max_queue_length, size_t inference_batch_size,
                       size_t num_episode_threads, size_t num_arena_games, size_t num_iters, size_t num_eps, size_t num_sims,
                       size_t num_sim_threads, size_t max_depth, int32_t cpuct) {
        if (inference_batch_size == 0 || num_sims % inference_batch_size != 0)
            throw Panic("assertion failed: num_sims % inference_batch_size == 0");        // src/coach.rs:83
        if (num_sim_threads == 0 || num_sims % num_sim_threads != 0)
            throw Panic("assertion failed: self.num_sims % self.num_threads == 0");             // src/async_mcts.rs:192
        Coach c(e);
        c.num_sim_threads = num_sim_threads;
        c.dir_ = checkpoint_directory;
        c.mcts_reserve_size = mcts_reserve_size; c.update_threshold = update_threshold; c.temp_threshold = temp_threshold;
        c.max_history_length = max_history_length; c.max_queue_length = max_queue_length;
        c.num_episode_threads = num_episode_threads; c.num_arena_games = num_arena_games; c.num_iters = num_iters;
        c.num_eps = num_eps; c.num_sims = num_sims; c.max_depth = max_depth; c.cpuct = cpuct;
        namespace fs = std::filesystem;
        fs::create_directories(c.dir_);
        long best = -1;                                        // resume from the largest numeric stem (:55-81)
        for (auto& ent : fs::directory_iterator(c.dir_)) {
            const std::string stem = ent.path().stem().string();
            if (ent.path().extension() != ".examples" || stem.empty() || stem.find_first_not_of("0123456789") != std::string::npos) continue;
            best = std::max(best, std::stol(stem));
        }
        if (best >= 0) {
            c.load_train_examples((size_t)best);
            c.start_iteration = (size_t)best + 1;
            // <dir>/coach.state = "iteration model_id": the accepted model after that iteration's gate; load it so the
            // restarted run continues from the live weights
            long it = -1, mid = -1;
            if (FILE* f = std::fopen((c.dir_ + "/coach.state").c_str(), "r")) {
                if (std::fscanf(f, "%ld %ld", &it, &mid) != 2) it = -1;
                std::fclose(f);
            }
            if (it == best && mid >= 0) {
                c.model_id = (size_t)mid;
                const std::string w = c.dir_ + "/" + std::to_string(mid) + ".aznet";
                if (fs::exists(w)) e.check(az_net_load(e.raw(), (int32_t)mid, w.c_str()));
            }
        }
        return c;
    }

    std::string examples_path(size_t iteration) const { return dir_ + "/" + std::to_string(iteration) + ".examples"; }

    void save_train_examples(size_t iteration) const {          // src/coach.rs:159-167
        FILE* f = std::fopen(examples_path(iteration).c_str(), "wb");
        if (!f) throw Panic("cannot write " + examples_path(iteration));
        const int64_t H = (int64_t)history.size();
        std::fwrite("AZEX0001", 1, 8, f);
        std::fwrite(&H, sizeof H, 1, f);
        for (auto& h : history) { const int64_t n = (int64_t)h.len(); std::fwrite(&n, sizeof n, 1, f); }
        for (auto& h : history) std::fwrite(h.boards.data(), sizeof(float), h.boards.size(), f);
        for (auto& h : history) std::fwrite(h.pis.data(), sizeof(float), h.pis.size(), f);
        for (auto& h : history) std::fwrite(h.vs.data(), sizeof(float), h.vs.size(), f);
        std::fclose(f);
    }

    void load_train_examples(size_t iteration) {
        FILE* f = std::fopen(examples_path(iteration).c_str(), "rb");
        if (!f) throw Panic("cannot read " + examples_path(iteration));
        char magic[8];
        int64_t H = 0;
        bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "AZEX0001", 8) == 0 && std::fread(&H, sizeof H, 1, f) == 1 && H >= 0;
        std::vector<int64_t> lens((size_t)(ok ? H : 0));
        ok = ok && std::fread(lens.data(), sizeof(int64_t), lens.size(), f) == lens.size();
        history.clear();
        for (int64_t n : lens) { HistoryEntry h; h.boards.resize((size_t)n * 84); h.pis.resize((size_t)n * 7); h.vs.resize((size_t)n); history.push_back(std::move(h)); }
        for (auto& h : history) ok = ok && std::fread(h.boards.data(), sizeof(float), h.boards.size(), f) == h.boards.size();
        for (auto& h : history) ok = ok && std::fread(h.pis.data(), sizeof(float), h.pis.size(), f) == h.pis.size();
        for (auto& h : history) ok = ok && std::fread(h.vs.data(), sizeof(float), h.vs.size(), f) == h.vs.size();
        std::fclose(f);
        if (!ok) throw Panic("malformed " + examples_path(iteration));
    }

    // One process per GPU: rank r of `world` plays the episode ids shard_range(num_eps, r, world) of every
    // iteration and the arena games shard_range(num_arena_games, r, world); the tuples meet in ONE az_gather_samples (every
    // rank receives: the trainer is replicated) and the arena tally in one 3-counter all-reduce.  world > 1 needs a
    // communicator on the engine (az_comm_unique_id on one rank, the 128 bytes shipped by the host, az_comm_init on all).
    void shard(int rank, int world) {
        if (world < 1 || rank < 0 || rank >= world) throw Panic("shard: rank outside the world");
        rank_ = rank; world_ = world;
    }
    static std::pair<size_t, size_t> shard_range(size_t n, int rank, int world) {
        return {n * (size_t)rank / (size_t)world, n * (size_t)(rank + 1) / (size_t)world};
    }

    // the self-play fan-out of src/coach.rs:241-272: num_eps x execute_episode, emitted with both symmetries
    HistoryEntry execute_episodes(size_t model_id, size_t iteration, uint64_t seed) {
        if (world_ > 1 || use_comm_at_world_1) return execute_episodes_sharded(model_id, iteration, seed);
        az_selfplay_params p{};
        p.n_games = (int32_t)num_eps; p.concurrent = (int32_t)std::min(num_episode_threads, num_eps);
        p.num_sims = (int32_t)num_sims; p.temp_threshold = (int32_t)temp_threshold; p.max_depth = (int32_t)max_depth;
        p.cpuct = cpuct; p.model_id = (int32_t)model_id; p.symmetries = 1; p.reserve = mcts_reserve_size; p.seed = seed;
        p.first_game_id = (uint64_t)(iteration * num_eps);
        p.num_sim_threads = (int32_t)num_sim_threads;
        const size_t cap = num_eps * 42 * 2;
        HistoryEntry h;
        h.boards.resize(cap * 84); h.pis.resize(cap * 7); h.vs.resize(cap);
        az_samples out{};
        out.capacity = (int64_t)cap; out.boards = h.boards.data(); out.pis = h.pis.data(); out.zs = h.vs.data();
        e_.check(az_selfplay(e_.raw(), &p, &out));
        const size_t n = (size_t)out.count;
        h.boards.resize(n * 84); h.pis.resize(n * 7); h.vs.resize(n);
        return h;
    }

    // the same fan-out over the ranks: this rank's shard of the episode ids as compact (state, pi, z) tuples, ONE gather in which
    // every rank receives all of them (rank order = episode-id order), symmetries regenerated here
    HistoryEntry execute_episodes_sharded(size_t model_id, size_t iteration, uint64_t seed) {
        const auto [lo, hi] = shard_range(num_eps, rank_, world_);
        const size_t mine = hi - lo;
        std::vector<uint64_t> st(std::max<size_t>(mine, 1) * 42 * 2);
        std::vector<float> pi(std::max<size_t>(mine, 1) * 42 * 7), z(std::max<size_t>(mine, 1) * 42);
        az_samples loc{};
        loc.capacity = (int64_t)(mine * 42); loc.states = st.data(); loc.pis = pi.data(); loc.zs = z.data();
        if (mine > 0) {
            az_selfplay_params p{};
            p.n_games = (int32_t)mine; p.concurrent = (int32_t)std::min(num_episode_threads, mine);
            p.num_sims = (int32_t)num_sims; p.temp_threshold = (int32_t)temp_threshold; p.max_depth = (int32_t)max_depth;
            p.cpuct = cpuct; p.model_id = (int32_t)model_id; p.symmetries = 0; p.reserve = mcts_reserve_size; p.seed = seed;
            p.first_game_id = (uint64_t)(iteration * num_eps + lo);
            p.num_sim_threads = (int32_t)num_sim_threads;
            e_.check(az_selfplay(e_.raw(), &p, &loc));
        }
        const size_t cap = num_eps * 42;
        std::vector<uint64_t> gs(cap * 2);
        std::vector<float> gp(cap * 7), gz(cap);
        az_samples all{};
        all.capacity = (int64_t)cap; all.states = gs.data(); all.pis = gp.data(); all.zs = gz.data();
        e_.check(az_gather_samples(e_.raw(), &loc, -1, &all, nullptr));
        const size_t n = (size_t)all.count;
        HistoryEntry h;
        h.boards.assign(n * 2 * 84, 0.f); h.pis.resize(n * 2 * 7); h.vs.resize(n * 2);
        for (size_t i = 0; i < n; ++i) {                         // get_symmetries (connect_four_game.rs:205-211): identity, then the mirror
            ConnectFourGame g;
            g.plus = gs[2 * i]; g.minus = gs[2 * i + 1];
            const Policy p(gp.begin() + (std::ptrdiff_t)(i * 7), gp.begin() + (std::ptrdiff_t)(i * 7 + 7));
            size_t k = 2 * i;
            for (auto& bp : g.get_symmetries(p)) {
                const BoardFeatures f = bp.first.to_features();
                std::copy(f.begin(), f.end(), h.boards.begin() + (std::ptrdiff_t)(k * 84));
                std::copy(bp.second.begin(), bp.second.end(), h.pis.begin() + (std::ptrdiff_t)(k * 7));
                h.vs[k] = gz[i];
                ++k;
            }
        }
        return h;
    }

    // Coach::learn(skip_first_play), src/coach.rs:169-396.  Engine model slots: `model_id` is the current net,
    // `model_id + 1` the candidate.
    static constexpr size_t RESUMED = (size_t)-1;     // learn(): continue from the live model of the resumed run (else 0)
    std::vector<Report> learn(bool skip_first_play, uint64_t seed, size_t model_id = RESUMED) {
        std::vector<Report> report;
        if (model_id == RESUMED) model_id = this->model_id;
        {   // the run's initial model: what a restart would load
            const std::string w = dir_ + "/" + std::to_string(model_id) + ".aznet";
            if (rank_ == 0 && !std::filesystem::exists(w)) e_.check(az_net_save(e_.raw(), (int32_t)model_id, w.c_str()));
        }
        for (size_t iteration = start_iteration; iteration < start_iteration + num_iters; ++iteration) {
            HistoryEntry h;
            if (!skip_first_play || iteration > start_iteration) {
                h = execute_episodes(model_id, iteration, seed);
                if (h.len() > max_queue_length) {                   // keep the newest max_queue_length (:275-277)
                    const size_t drop = h.len() - max_queue_length;
                    h.boards.erase(h.boards.begin(), h.boards.begin() + (std::ptrdiff_t)(drop * 84));
                    h.pis.erase(h.pis.begin(), h.pis.begin() + (std::ptrdiff_t)(drop * 7));
                    h.vs.erase(h.vs.begin(), h.vs.begin() + (std::ptrdiff_t)drop);
                }
            }
            history.push_back(std::move(h));                                    // :282: pushed even when the play was skipped
            if (history.size() > max_history_length) history.pop_front();      // :285-288
            if (rank_ == 0) save_train_examples(iteration);                       // :291-293 (one writer per run)
            size_t n = 0;
            for (auto& h : history) n += h.len();
            if (n == 0) throw Panic("assertion failed: training set is empty");   // :305
            std::vector<float> ab, ap, av;
            ab.reserve(n * 84); ap.reserve(n * 7); av.reserve(n);
            for (auto& h : history) { ab.insert(ab.end(), h.boards.begin(), h.boards.end()); ap.insert(ap.end(), h.pis.begin(), h.pis.end()); av.insert(av.end(), h.vs.begin(), h.vs.end()); }
            const std::vector<int64_t> perm = shuffle_permutation(n, seed, iteration);   // :296-297
            std::vector<float> sb(n * 84), sp(n * 7), sv(n);
            for (size_t i = 0; i < n; ++i) {
                const size_t src = (size_t)perm[i];
                std::memcpy(&sb[i * 84], &ab[src * 84], 84 * sizeof(float));
                std::memcpy(&sp[i * 7], &ap[src * 7], 7 * sizeof(float));
                sv[i] = av[src];
            }
            e_.check(az_set_option(e_.raw(), "train_seed", (int64_t)(seed + iteration)));
            e_.check(az_net_train(e_.raw(), (int32_t)model_id, (int32_t)model_id + 1, sb.data(), sp.data(), sv.data(), (int64_t)n));   // :329
            if (rank_ == 0) e_.check(az_net_save(e_.raw(), (int32_t)model_id + 1, (dir_ + "/" + std::to_string(model_id + 1) + ".aznet").c_str()));
            Report r{};
            r.iteration = iteration; r.samples = n; r.model_id = model_id;
            r.losses.resize(2 * (size_t)az_net_train_history(e_.raw(), nullptr, 0));
            az_net_train_history(e_.raw(), r.losses.data(), (int32_t)(r.losses.size() / 2));
            // arena: new (first listed) vs old, both seatings (:333-375)
            az_arena_params a{};
            a.num_games = (int32_t)num_arena_games; a.num_sims = (int32_t)num_sims; a.max_depth = (int32_t)max_depth; a.cpuct = cpuct;
            a.new_model_id = (int32_t)model_id + 1; a.old_model_id = (int32_t)model_id; a.reserve = mcts_reserve_size;
            a.seed = seed + 7919ull * (uint64_t)(iteration + 1);
            a.num_sim_threads = (int32_t)num_sim_threads;
            if (world_ > 1 || use_comm_at_world_1) {                 // this rank's games of the arena, one 3-counter all-reduce
                const size_t total = 2 * (num_arena_games / 2);      // num/2 per seating (src/arena.rs:83)
                const auto [alo, ahi] = shard_range(total, rank_, world_);
                a.first_game = (int32_t)alo; a.num_games = (int32_t)(ahi - alo); a.total_games = (int32_t)total; a.allreduce_wld = 1;
                if (total == 0) { a.total_games = 0; a.first_game = 0; a.num_games = 0; a.allreduce_wld = 0; }
            }
            uint64_t wld[3] = {0, 0, 0};
            e_.check(az_arena(e_.raw(), &a, wld, nullptr));
            r.nwins = (size_t)wld[0]; r.pwins = (size_t)wld[1]; r.draws = (size_t)wld[2];
            std::printf("NEW/PREV WINS : %zu / %zu; DRAWS : %zu\n", r.nwins, r.pwins, r.draws);        // :381
            r.accepted = !(r.pwins + r.nwins == 0 || (float)r.nwins / (float)(r.pwins + r.nwins) < update_threshold);   // :383-390
            std::printf(r.accepted ? "ACCEPTING NEW MODEL\n" : "REJECTING NEW MODEL\n");
            // a long run moves to a new model id per accepted iteration: drop the slot nobody will read again
            e_.check(az_net_free(e_.raw(), (int32_t)(r.accepted ? model_id : model_id + 1)));
            if (r.accepted) ++model_id;
            if (rank_ == 0) {
                const std::string tmp = dir_ + "/coach.state.tmp";
                if (FILE* f = std::fopen(tmp.c_str(), "w")) {
                    std::fprintf(f, "%zu %zu\n", iteration, model_id);
                    std::fclose(f);
                    std::filesystem::rename(tmp, dir_ + "/coach.state");
                }
            }
            this->model_id = model_id;
            report.push_back(std::move(r));
        }
        return report;
    }

    std::deque<HistoryEntry> history;
    size_t start_iteration = 0, model_id = 0;
    size_t mcts_reserve_size = 0, temp_threshold = 0, max_history_length = 0, max_queue_length = 0, num_episode_threads = 0,
           num_arena_games = 0, num_iters = 0, num_eps = 0, num_sims = 0, max_depth = 0, num_sim_threads = 1;
    bool use_comm_at_world_1 = false;     // tests: run the gather / all-reduce path at world size 1
    float update_threshold = 0.f;
    int32_t cpuct = 1;

  private:
    explicit Coach(Engine& e) : e_(e) {}
    Engine& e_;
    std::string dir_;
    int rank_ = 0, world_ = 1;
};

}  // namespace az_host
