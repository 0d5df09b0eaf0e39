// az_oracle_capi.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
//
// extern "C" surface over az_oracle.hpp / az_oracle_games.hpp so that tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive the oracle
// through ctypes.  Nothing in the shipped engine links this file.
#include "az_oracle_games.hpp"

#include <thread>

using namespace azo;

namespace {

enum NetKind { NET_STUB = 0, NET_HASH = 1, NET_REPLAY = 2, NET_CALLBACK = 3 };

// NNet::predict supplied by the caller (bench.py's cpu_baseline leg: the f32 conv net on torch-CPU).
typedef void (*azo_predict_cb)(const float* boards, int B, int model_id, float* pi, float* v);
azo_predict_cb g_predict_cb = nullptr;
struct CallbackNet : NNet {
    void predict(const float* boards, int B, int model_id, float* pi, float* v) override {
        if (!g_predict_cb) throw std::runtime_error("no predict callback set");
        g_predict_cb(boards, B, model_id, pi, v);
    }
};

struct TreeBase {
    virtual ~TreeBase() = default;
    virtual int debug_children(uint64_t mine, uint64_t theirs, const int* path, int depth, uint64_t* out) = 0;
    virtual void set_sims(size_t n) = 0;
    virtual int get_action_prob(uint64_t mine, uint64_t theirs, float temp, uint64_t seed, uint64_t game_id,
                                float* pi, uint16_t* counts, float* q) = 0;
    virtual SearchStats stats() = 0;
    virtual size_t n_nodes() = 0;
    virtual ReplayNet* replay_net() = 0;
};

C4Array array_from_bits(uint64_t mine, uint64_t theirs, bool literal) {
    C4Array g;
    g.literal_windows = literal;
    for (int c = 0; c < C4_W; ++c)
        for (int row = 0; row < C4_H; ++row) {
            uint64_t bit = 1ull << (c * 7 + row);
            int r = C4_H - 1 - row;
            if (mine & bit) { g.s[r][c] = 1; g.heights[c] = (uint8_t)(row + 1); }
            else if (theirs & bit) { g.s[r][c] = -1; g.heights[c] = (uint8_t)(row + 1); }
        }
    return g;
}

template <class G> G make_state(uint64_t mine, uint64_t theirs, const Quirks& q);
template <> C4Bits make_state<C4Bits>(uint64_t mine, uint64_t theirs, const Quirks&) { return C4Bits{mine, theirs}; }
template <> C3Bits make_state<C3Bits>(uint64_t mine, uint64_t theirs, const Quirks&) { return C3Bits{mine, theirs}; }
template <> C4Array make_state<C4Array>(uint64_t mine, uint64_t theirs, const Quirks& q) {
    return array_from_bits(mine, theirs, q.b6_literal_windows);
}

struct NetBox {
    StubNet stub;
    HashNet hash;
    ReplayNet replay;
    CallbackNet callback;
    NNet* get(int kind) {
        if (kind == NET_STUB) return &stub;
        if (kind == NET_HASH) return &hash;
        if (kind == NET_CALLBACK) return &callback;
        return &replay;
    }
};

template <class G>
struct TreeImpl : TreeBase {
    NetBox nets;
    std::unique_ptr<AsyncMcts<G>> mcts;
    Quirks quirks;
    // eval records (for replay parity the other way round and for debugging)
    std::vector<uint64_t> rec_states;
    std::vector<float> rec_pi, rec_v;

    int get_action_prob(uint64_t mine, uint64_t theirs, float temp, uint64_t seed, uint64_t game_id, float* pi,
                        uint16_t* counts, float* q) override {
        try {
            G s = make_state<G>(mine, theirs, quirks);
            uint64_t ply = (uint64_t)__builtin_popcountll(mine | theirs);
            auto p = mcts->get_action_prob(s, temp, seed, game_id, ply, counts, q);
            for (size_t i = 0; i < p.size(); ++i) pi[i] = p[i];
            return 0;
        } catch (const std::exception&) {
            return -1;
        }
    }
    // diagnostics: children of the node reached from the node of state (mine, theirs) by following `path` (child indices);
    // rows of 8 u64: slot, a, ctr (resolved), prior bits, link (or ~0), expanded, own ctr, 0; row 7: the node's own ctr
    int debug_children(uint64_t mine, uint64_t theirs, const int* path, int depth, uint64_t* out) override {
        auto root = mcts->nodes->lookup_state_id(make_state<G>(mine, theirs, quirks));
        if (!root) return -1;
        size_t cur = *root;
        for (int i = 0; i < depth; ++i) cur = *mcts->nodes->resolve(mcts->nodes->get(cur)->children[(size_t)path[i]]);
        Node<G>* nd = mcts->nodes->get(cur);
        int j = 0;
        for (size_t ci : nd->children) {
            Node<G>* ch = mcts->nodes->get(ci);
            uint8_t a = mcts->nodes->raw(ci)->a;
            float p = nd->p ? (*nd->p)[a] : 0.0f;
            uint32_t pb;
            std::memcpy(&pb, &p, 4);
            uint64_t* o = out + 8 * j;
            o[0] = ci; o[1] = a; o[2] = ch->win_counter.load(); o[3] = pb;
            o[4] = mcts->nodes->buf[ci].link ? *mcts->nodes->buf[ci].link : ~0ull; o[5] = ch->s ? 1 : 0; o[6] = mcts->nodes->raw(ci)->win_counter.load(); o[7] = 0;
            ++j;
        }
        out[8 * 7] = nd->win_counter.load(); out[8 * 7 + 2] = cur;
        return j;
    }
    void set_sims(size_t n) override { mcts->num_sims = n; }
    SearchStats stats() override { return mcts->stats; }
    size_t n_nodes() override { return mcts->nodes->size(); }
    ReplayNet* replay_net() override { return &nets.replay; }
};

Quirks quirks_from_bits(uint32_t b) {
    Quirks q;
    q.b1_parent_action = b & 1;
    q.b2_same_sign_backup = b & 2;
    q.b4_literal_z = b & 4;
    q.b6_literal_windows = b & 8;
    return q;
}

template <class G>
TreeBase* make_tree(int has_root, uint64_t mine, uint64_t theirs, size_t reserve, size_t sims, size_t max_depth,
                    size_t model_id, int cpuct, int net_kind, uint64_t salt, uint32_t qbits, size_t threads = 1, bool force_lockstep = false) {
    auto* t = new TreeImpl<G>();
    t->quirks = quirks_from_bits(qbits);
    t->nets.hash.salt = salt;
    NNet* net = t->nets.get(net_kind);
    if (has_root)
        t->mcts.reset(new AsyncMcts<G>(make_state<G>(mine, theirs, t->quirks), reserve, sims, threads, max_depth,
                                       model_id, cpuct, net, C4_W));
    else
        t->mcts.reset(new AsyncMcts<G>(reserve, sims, threads, max_depth, model_id, cpuct, net, C4_W));
    t->mcts->quirks = t->quirks;
    t->mcts->force_lockstep = force_lockstep;
    return t;
}

}  // namespace

extern "C" {

void azo_set_predict_callback(azo_predict_cb cb) { g_predict_cb = cb; }

// ---- packed counter, src/node.rs:16-93 -----------------------------------
uint64_t azo_ctr_init() { return 0x7FFFFFFF00000000ull; }
uint64_t azo_ctr_visit(uint64_t c) {
    Node<DummyGame> n; n.win_counter.store(c); n.visit(); return n.win_counter.load();
}
uint64_t azo_ctr_unvisit(uint64_t c, float v, float scale) {
    Node<DummyGame> n(scale); n.win_counter.store(c); n.unvisit(v); return n.win_counter.load();
}
float azo_ctr_w(uint64_t c, float scale) { Node<DummyGame> n(scale); n.win_counter.store(c); return n.get_w(); }
uint32_t azo_ctr_n(uint64_t c) { Node<DummyGame> n; n.win_counter.store(c); return n.get_n(); }
uint32_t azo_ctr_vloss(uint64_t c) { Node<DummyGame> n; n.win_counter.store(c); return n.get_vloss(); }
float azo_ctr_q(uint64_t c, float scale) { Node<DummyGame> n(scale); n.win_counter.store(c); return n.compute_q(); }
// PUCT term of best_child, src/node.rs:352-356 (C6)
float azo_puct(uint64_t child_ctr, float prior, uint32_t parent_n, int32_t cpuct) {
    Node<DummyGame> ch; ch.win_counter.store(child_ctr);
    return ch.compute_q() +
           (((float)cpuct * prior) * std::sqrt((float)(uint16_t)parent_n + EPS)) / (float)(uint16_t)(1 + ch.get_n());
}

// ---- RNG of the build -------------------------------------------------------
uint64_t azo_rng_draw(uint64_t seed, uint64_t game_id, uint64_t ply, uint64_t purpose) {
    return rng_draw(seed, game_id, ply, purpose);
}
uint32_t azo_rng_choose(uint64_t r, uint32_t k) { return rng_choose(r, k); }
int azo_rng_choose_weighted(uint64_t r, const float* w, int n) { return rng_choose_weighted(r, w, n); }

// ---- Connect Four on canonical (mine, theirs) bitboards --------------------
// canonical successor: play `a` for the side to move, then swap sides
void azo_c4_play(uint64_t mine, uint64_t theirs, int a, uint64_t* out2) {
    C4Bits b{mine, theirs};
    auto nx = b.get_next_state(1, (uint8_t)a);
    C4Bits c = nx.first.get_canonical_form(nx.second);
    out2[0] = c.p1; out2[1] = c.m1;
}
float azo_c4_ended(uint64_t mine, uint64_t theirs) { return C4Bits{mine, theirs}.get_game_ended(1); }
float azo_c3_ended(uint64_t mine, uint64_t theirs) { return C3Bits{mine, theirs}.get_game_ended(1); }
float azo_c4_ended_array(uint64_t mine, uint64_t theirs, int literal) {
    return array_from_bits(mine, theirs, literal != 0).get_game_ended(1);
}
int azo_c4_valid_mask(uint64_t mine, uint64_t theirs) {
    auto v = C4Bits{mine, theirs}.get_valid_moves(1);
    int m = 0;
    for (int c = 0; c < C4_W; ++c) if (v[c]) m |= 1 << c;
    return m;
}
void azo_c4_features(uint64_t mine, uint64_t theirs, float* out84) {
    auto f = C4Bits{mine, theirs}.to_features();
    std::memcpy(out84, f.data(), 84 * sizeof(float));
}
void azo_c4_features_array(uint64_t mine, uint64_t theirs, float* out84) {
    auto f = array_from_bits(mine, theirs, false).to_features();
    std::memcpy(out84, f.data(), 84 * sizeof(float));
}
uint64_t azo_c4_mirror(uint64_t b) { return C4Bits::mirror(b); }
void azo_hashnet(uint64_t mine, uint64_t theirs, uint64_t salt, float* pi7, float* v) {
    hashnet_eval(mine, theirs, salt, pi7, v);
}

// ---- AsyncMcts -------------------------------------------------------------
void* azo_tree_new(int game_kind, int has_root, uint64_t mine, uint64_t theirs, uint64_t reserve, uint64_t sims,
                   uint64_t max_depth, uint64_t model_id, int cpuct, int net_kind, uint64_t salt, uint32_t qbits) {
    try {
        if (game_kind == 0)
            return make_tree<C4Bits>(has_root, mine, theirs, reserve, sims, max_depth, model_id, cpuct, net_kind, salt, qbits);
        if (game_kind == 2)
            return make_tree<C3Bits>(has_root, mine, theirs, reserve, sims, max_depth, model_id, cpuct, net_kind, salt, qbits);
        return make_tree<C4Array>(has_root, mine, theirs, reserve, sims, max_depth, model_id, cpuct, net_kind, salt, qbits);
    } catch (const std::exception&) {
        return nullptr;
    }
}
// the same with num_threads simulations in flight per tree (lock-step schedule, az_oracle.hpp); force_lockstep runs
// num_threads == 1 through the lock-step code as well
void* azo_tree_new_mt(int game_kind, int has_root, uint64_t mine, uint64_t theirs, uint64_t reserve, uint64_t sims, uint64_t threads,
                      uint64_t max_depth, uint64_t model_id, int cpuct, int net_kind, uint64_t salt, uint32_t qbits, int force_lockstep) {
    try {
        if (game_kind == 0)
            return make_tree<C4Bits>(has_root, mine, theirs, reserve, sims, max_depth, model_id, cpuct, net_kind, salt, qbits, threads, force_lockstep != 0);
        if (game_kind == 2)
            return make_tree<C3Bits>(has_root, mine, theirs, reserve, sims, max_depth, model_id, cpuct, net_kind, salt, qbits, threads, force_lockstep != 0);
        return make_tree<C4Array>(has_root, mine, theirs, reserve, sims, max_depth, model_id, cpuct, net_kind, salt, qbits, threads, force_lockstep != 0);
    } catch (const std::exception&) {
        return nullptr;
    }
}
int azo_tree_debug_children(void* t, uint64_t mine, uint64_t theirs, const int* path, int depth, uint64_t* out) {
    return ((TreeBase*)t)->debug_children(mine, theirs, path, depth, out);
}
void azo_tree_set_sims(void* t, uint64_t n) { ((TreeBase*)t)->set_sims((size_t)n); }
void azo_tree_free(void* t) { delete (TreeBase*)t; }
int azo_tree_get_action_prob(void* t, uint64_t mine, uint64_t theirs, float temp, uint64_t seed, uint64_t game_id,
                             float* pi, uint16_t* counts, float* q) {
    return ((TreeBase*)t)->get_action_prob(mine, theirs, temp, seed, game_id, pi, counts, q);
}
// replay records for a tree created with net_kind 2 (borrowed pointers; must outlive the calls)
void azo_tree_set_replay(void* t, const uint64_t* states, const float* pis, const float* vs, uint64_t n) {
    ReplayNet* r = ((TreeBase*)t)->replay_net();
    r->states = states; r->pis = pis; r->vs = vs; r->n = (size_t)n; r->pos = 0; r->mismatch = false;
}
int azo_tree_replay_bad(void* t) { return ((TreeBase*)t)->replay_net()->mismatch ? 1 : 0; }
void azo_tree_stats(void* t, uint64_t* out8) {
    SearchStats s = ((TreeBase*)t)->stats();
    out8[0] = s.sims; out8[1] = s.expansions; out8[2] = s.leaf_evals; out8[3] = s.link_hits;
    out8[4] = s.terminal_hits; out8[5] = s.depth_sum; out8[6] = ((TreeBase*)t)->n_nodes(); out8[7] = s.abandoned;
}

// ---- Coach::execute_episode x n_games ---------------------------------------
// Outputs: boards [cap,2,6,7], pis [cap,7], zs [cap] (both symmetries per ply,
// game-id order then ply order), game_len [n_games] plies, moves [n_games,42],
// stats [6] summed over games.  Replay mode (net_kind 2): per-game record
// ranges rec_off[n_games+1] into rec_states/rec_pi/rec_v; replay_bad[g] != 0
// if the search asked for a different state or more records than were given.
// Returns number of samples written, or -1 on error / overflow of cap.
int64_t azo_selfplay(int64_t n_games, uint64_t first_game_id, uint64_t sims, uint64_t temp_threshold, int cpuct,
                     uint64_t max_depth, uint64_t reserve, uint64_t seed, int net_kind, uint64_t salt,
                     int game_kind, uint32_t qbits, int threads, float* boards, float* pis, float* zs,
                     int64_t cap, int32_t* game_len, uint8_t* moves, uint64_t* stats6, const int64_t* rec_off,
                     const uint64_t* rec_states, const float* rec_pi, const float* rec_v, int32_t* replay_bad, int sim_threads) {
    const size_t ST = sim_threads > 0 ? (size_t)sim_threads : 1;     // num_sim_threads, src/coach.rs:249
    struct PerGame { std::vector<TrainingSample> s; std::vector<uint8_t> moves; SearchStats st; bool bad = false; bool err = false; };
    std::vector<PerGame> res((size_t)n_games);
    auto run_one = [&](int64_t g) {
        try {
            NetBox nets;
            nets.hash.salt = salt;
            if (net_kind == NET_REPLAY) {
                nets.replay.states = rec_states ? rec_states + 2 * rec_off[g] : nullptr;
                nets.replay.pis = rec_pi + 7 * rec_off[g];
                nets.replay.vs = rec_v + rec_off[g];
                nets.replay.n = (size_t)(rec_off[g + 1] - rec_off[g]);
            }
            Quirks q = quirks_from_bits(qbits);
            if (game_kind == 0) {
                AsyncMcts<C4Bits> m(reserve, sims, ST, max_depth, 0, cpuct, nets.get(net_kind), C4_W);
                m.quirks = q;
                res[g].s = execute_episode<C4Bits>(m, temp_threshold, seed, first_game_id + (uint64_t)g, &res[g].moves);
                res[g].st = m.stats;
            } else if (game_kind == 2) {
                AsyncMcts<C3Bits> m(reserve, sims, ST, max_depth, 0, cpuct, nets.get(net_kind), C4_W);
                m.quirks = q;
                res[g].s = execute_episode<C3Bits>(m, temp_threshold, seed, first_game_id + (uint64_t)g, &res[g].moves);
                res[g].st = m.stats;
            } else {
                AsyncMcts<C4Array> m(reserve, sims, ST, max_depth, 0, cpuct, nets.get(net_kind), C4_W);
                m.quirks = q;
                res[g].s = execute_episode<C4Array>(m, temp_threshold, seed, first_game_id + (uint64_t)g, &res[g].moves);
                res[g].st = m.stats;
            }
            if (net_kind == NET_REPLAY) res[g].bad = nets.replay.mismatch || nets.replay.pos != nets.replay.n;
        } catch (const std::exception&) {
            res[g].err = true;
        }
    };
    if (threads <= 1) {
        for (int64_t g = 0; g < n_games; ++g) run_one(g);
    } else {
        std::atomic<int64_t> next{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&] { for (int64_t g; (g = next.fetch_add(1)) < n_games;) run_one(g); });
        for (auto& th : pool) th.join();
    }
    int64_t n = 0;
    SearchStats tot;
    for (int64_t g = 0; g < n_games; ++g) {
        if (res[g].err) return -1;
        if (game_len) game_len[g] = (int32_t)res[g].moves.size();
        if (moves) for (size_t i = 0; i < res[g].moves.size() && i < 42; ++i) moves[g * 42 + i] = res[g].moves[i];
        if (replay_bad) replay_bad[g] = res[g].bad ? 1 : 0;
        tot.sims += res[g].st.sims; tot.expansions += res[g].st.expansions; tot.leaf_evals += res[g].st.leaf_evals;
        tot.link_hits += res[g].st.link_hits; tot.terminal_hits += res[g].st.terminal_hits; tot.depth_sum += res[g].st.depth_sum;
        for (auto& ts : res[g].s) {
            if (boards) {
                if (n >= cap) return -1;
                std::memcpy(boards + n * 84, ts.board.data(), 84 * sizeof(float));
                std::memcpy(pis + n * 7, ts.pi.data(), 7 * sizeof(float));
                zs[n] = ts.v;
            }
            ++n;
        }
    }
    if (stats6) { stats6[0] = tot.sims; stats6[1] = tot.expansions; stats6[2] = tot.leaf_evals; stats6[3] = tot.link_hits; stats6[4] = tot.terminal_hits; stats6[5] = tot.depth_sum; }
    return n;
}

// ---- arena::play_games (C16), per-game tree pair (B8) -----------------------
// model slot 0 = new net (salt_new / model id 1), slot 1 = old net (model id 0).
// results[g] (optional, [num]) = +1 first seat won, -1 second seat won, 0 draw.
}  // extern "C" (the arena body is a template over the game; its C entry points follow it)

static uint64_t g_last_bad_req[2] = {0, 0};      // diagnostics: the state requested at the first bad replay record of the last arena
extern "C" void azo_debug_last_bad_request(uint64_t* out2) { out2[0] = g_last_bad_req[0]; out2[1] = g_last_bad_req[1]; }

// Replay streams of an arena (replay parity of az_arena with recorded conv-net rows): per game and per model the
// (state, pi, v) rows that model's tree consumed, flattened with offsets [n_games + 1]
struct ArenaReplay {
    const int64_t* off[2] = {nullptr, nullptr};        // [0] new model, [1] old model
    const uint64_t* states[2] = {nullptr, nullptr};
    const float* pi[2] = {nullptr, nullptr};
    const float* v[2] = {nullptr, nullptr};
    int32_t* bad = nullptr;                            // [n_games]
};

// games [first_game, first_game + n_games) of an arena of `total` games (seating and RNG by the global game index);
// start = play_games' `board: Option<G>` (src/arena.rs:62-67), nullptr = None
template <class G>
static int arena_impl(uint64_t total, uint64_t first_game, uint64_t n_games, uint64_t sims, size_t sim_threads, int cpuct, uint64_t max_depth,
                      uint64_t reserve, uint64_t seed, int net_kind, uint64_t salt, int new_model_id, int old_model_id, int threads,
                      uint64_t* wld3, int8_t* results, const uint64_t* start, const ArenaReplay* rp) {
    try {
        const uint64_t half = total / 2;
        std::vector<int8_t> res(n_games, 0);
        std::optional<G> board0;
        if (start) board0 = G{start[0], start[1]};
        auto run_one = [&](uint64_t li) {
            const uint64_t gi = first_game + li;
            int first = gi < half ? 0 : 1;
            NetBox nets[2];                             // [0] the new model's net, [1] the old model's
            for (int m = 0; m < 2; ++m) {
                nets[m].hash.salt = salt;
                if (net_kind == NET_REPLAY && rp) {
                    nets[m].replay.states = rp->states[m] ? rp->states[m] + 2 * rp->off[m][li] : nullptr;
                    nets[m].replay.pis = rp->pi[m] + 7 * rp->off[m][li];
                    nets[m].replay.vs = rp->v[m] + rp->off[m][li];
                    nets[m].replay.n = (size_t)(rp->off[m][li + 1] - rp->off[m][li]);
                }
            }
            AsyncMcts<G> trees[2] = {
                AsyncMcts<G>(reserve, sims, sim_threads, max_depth, (size_t)new_model_id, cpuct, nets[0].get(net_kind), C4_W),
                AsyncMcts<G>(reserve, sims, sim_threads, max_depth, (size_t)old_model_id, cpuct, nets[1].get(net_kind), C4_W)};
            auto mk = [&](int slot) {
                return std::function<uint8_t(const G&)>([&trees, slot, seed, gi](const G& s) {
                    uint64_t ply = (uint64_t)__builtin_popcountll(s.p1 | s.m1);
                    auto p = trees[slot].get_action_prob(s, 0.0f, seed, gi, ply);   // src/coach.rs:369-372
                    // argmax with max_by (last max), src/coach.rs:356-363
                    size_t best = 0;
                    for (size_t i = 1; i < p.size(); ++i) if (!(p[best] > p[i])) best = i;
                    return (uint8_t)best;
                });
            };
            std::function<uint8_t(const G&)> acts[2] = {mk(first), mk(1 - first)};
            res[li] = play_game<G>(acts, board0);
            if (net_kind == NET_REPLAY && rp && rp->bad) {
                // 0 = replayed cleanly; else bit 0 / 1: the new / old model's tree asked for a state its record does not hold,
                // bit 2 / 3: ... left records unused; bits 8..: index of the first bad request of the first failing tree
                int b = (nets[0].replay.mismatch ? 1 : 0) | (nets[1].replay.mismatch ? 2 : 0) | (nets[0].replay.pos != nets[0].replay.n ? 4 : 0) |
                        (nets[1].replay.pos != nets[1].replay.n ? 8 : 0);
                const long fb = nets[0].replay.first_bad >= 0 ? nets[0].replay.first_bad : nets[1].replay.first_bad;
                if (b && fb >= 0) b |= (int)(fb << 8);
                if (b && fb >= 0) { const ReplayNet& r = nets[0].replay.first_bad >= 0 ? nets[0].replay : nets[1].replay; g_last_bad_req[0] = r.bad_req[0]; g_last_bad_req[1] = r.bad_req[1]; }
                rp->bad[li] = b;
            }
        };
        if (threads <= 1) {
            for (uint64_t g = 0; g < n_games; ++g) run_one(g);
        } else {
            std::atomic<uint64_t> next{0};
            std::vector<std::thread> pool;
            std::atomic<bool> failed{false};
            for (int t = 0; t < threads; ++t)
                pool.emplace_back([&] {
                    try { for (uint64_t g; (g = next.fetch_add(1)) < n_games;) run_one(g); }
                    catch (...) { failed = true; }
                });
            for (auto& th : pool) th.join();
            if (failed) return -1;
        }
        wld3[0] = wld3[1] = wld3[2] = 0;
        for (uint64_t li = 0; li < n_games; ++li) {
            int first = first_game + li < half ? 0 : 1;
            int win_cond = first == 0 ? 1 : -1, lose_cond = first == 0 ? -1 : 1;   // src/arena.rs:80-81
            if (res[li] == win_cond) wld3[0]++;
            else if (res[li] == lose_cond) wld3[1]++;
            else wld3[2]++;
            if (results) results[li] = res[li];
        }
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

extern "C" {
int azo_arena(uint64_t num, uint64_t sims, int cpuct, uint64_t max_depth, uint64_t reserve, uint64_t seed, int net_kind,
              uint64_t salt, int new_model_id, int old_model_id, int threads, uint64_t* wld3, int8_t* results) {
    return arena_impl<CBits<4>>(2 * (num / 2), 0, 2 * (num / 2), sims, 1, cpuct, max_depth, reserve, seed, net_kind, salt, new_model_id, old_model_id, threads,
                                wld3, results, nullptr, nullptr);
}
int azo_arena_c3(uint64_t num, uint64_t sims, int cpuct, uint64_t max_depth, uint64_t reserve, uint64_t seed, int net_kind,
                 uint64_t salt, int new_model_id, int old_model_id, int threads, uint64_t* wld3, int8_t* results) {
    return arena_impl<CBits<3>>(2 * (num / 2), 0, 2 * (num / 2), sims, 1, cpuct, max_depth, reserve, seed, net_kind, salt, new_model_id, old_model_id, threads,
                                wld3, results, nullptr, nullptr);
}
// The general form: games [first_game, first_game + n_games) of a `total`-game arena (Connect Four), sim_threads simulations in
// flight per tree, an optional start board [2] and optional replay streams (net_kind 2): for model m (0 new, 1 old)
// off_m [n_games + 1] into states_m [N,2] (may be NULL) / pi_m [N,7] / v_m [N]; replay_bad [n_games].
int azo_arena_ex(uint64_t total, uint64_t first_game, uint64_t n_games, uint64_t sims, int sim_threads, int cpuct, uint64_t max_depth,
                 uint64_t reserve, uint64_t seed, int net_kind, uint64_t salt, int new_model_id, int old_model_id, int threads,
                 uint64_t* wld3, int8_t* results, const uint64_t* start_board,
                 const int64_t* off_new, const uint64_t* states_new, const float* pi_new, const float* v_new,
                 const int64_t* off_old, const uint64_t* states_old, const float* pi_old, const float* v_old, int32_t* replay_bad) {
    ArenaReplay rp;
    rp.off[0] = off_new; rp.states[0] = states_new; rp.pi[0] = pi_new; rp.v[0] = v_new;
    rp.off[1] = off_old; rp.states[1] = states_old; rp.pi[1] = pi_old; rp.v[1] = v_old;
    rp.bad = replay_bad;
    return arena_impl<CBits<4>>(total, first_game, n_games, sims, sim_threads > 0 ? (size_t)sim_threads : 1, cpuct, max_depth, reserve, seed, net_kind,
                                salt, new_model_id, old_model_id, threads, wld3, results, start_board, off_new ? &rp : nullptr);
}
}  // extern "C"
