// az_oracle.hpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT PATH).
//
// A plain C++17 restatement of the reference's async_mcts + arena hot path
// (AnimatedRNG/alphazero-rs).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may build, link or call anything in oracle/.
// The shipped engine (alphazero-rs_amd/csrc) never includes this file.
//
// PARITY STATUS
//   * Packed win counter, NodeStore state machine, Connect Four diagonal:
//     pinned by the reference's own known-answer tests
//     (src/node.rs:393-655, examples/connect_four_lib/connect_four_game.rs:244-264),
//     re-run in oracle/test_oracle.cpp.
//   * best_child / search / get_action_prob / play_game(s) / execute_episode:
//     PARITY UNPINNED BY THE REFERENCE.  The reference has no test for them,
//     cannot be compiled here (no Rust toolchain) and panics as written
//     (SURVEY.md section 0.2, A1-A12).  Those functions follow the reference
//     line by line with the repairs S1-S10 / B1-B10 of SURVEY.md section 0.2,
//     each marked at the place it applies.
//
// Every function cites the reference file:line it restates.
#pragma once
#include <array>
#include <atomic>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <unordered_map>
#include <utility>
#include <vector>

namespace azo {

// src/node.rs:12-13
constexpr float EPS = 1e-6f;
constexpr float WIN_SCALE = 100.0f;

// ---------------------------------------------------------------------------
// Repair toggles.  Default = repaired behaviour (SURVEY.md section 0.2 class B).
// Setting a flag restores the literal reference behaviour where that is a
// one-line difference, so the cost of each repair stays auditable.
struct Quirks {
    bool b1_parent_action = false;   // src/async_mcts.rs:280-284 plays node_p.a
    bool b2_same_sign_backup = false;// src/async_mcts.rs:361-370 no sign flip
    bool b4_literal_z = false;       // src/coach.rs:144-154 z ignores r
    bool b6_literal_windows = false; // connect_four_game.rs:114,129 short loops
};

// ---------------------------------------------------------------------------
// Counter-based RNG of the build (the reference clones one
// SmallRng state per episode / per arena move; rand 0.7.3 is not available and
// no reference test pins an RNG-dependent output, so the build defines its own
// stream keyed on (seed, global game id, ply, purpose)).  Shared bit-for-bit
// with the HIP engine (alphazero-rs_amd/csrc/az_common.h restates it).
inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
enum RngPurpose : uint64_t { RNG_TIEBREAK = 1, RNG_MOVE = 2 };
inline uint64_t rng_draw(uint64_t seed, uint64_t game_id, uint64_t ply, uint64_t purpose) {
    return mix64(mix64(mix64(mix64(seed) ^ game_id) ^ ply) ^ purpose);
}
// uniform pick among k candidates (stands in for IteratorRandom::choose,
// src/async_mcts.rs:99-105)
inline uint32_t rng_choose(uint64_t r, uint32_t k) {
    return (uint32_t)(((r >> 32) * (uint64_t)k) >> 32);
}
// index sampled proportionally to w (stands in for SliceRandom::choose_weighted,
// src/coach.rs:137-138).  24-bit uniform, sequential f32 cumulative walk.
inline int rng_choose_weighted(uint64_t r, const float* w, int n) {
    float total = 0.0f;
    for (int i = 0; i < n; ++i) total = total + w[i];
    float u = (float)(uint32_t)(r >> 40) * (1.0f / 16777216.0f);
    float t = u * total;
    float acc = 0.0f;
    int last = -1;
    for (int i = 0; i < n; ++i) {
        if (w[i] > 0.0f) {
            acc = acc + w[i];
            last = i;
            if (t < acc) return i;
        }
    }
    return last;
}

// ---------------------------------------------------------------------------
// Node: src/node.rs:16-93.  One packed 64-bit word 0xWWWWWWWW_NNNN_VVVV.
template <class G>
struct Node {
    std::atomic<uint64_t> win_counter{0x7FFFFFFF00000000ull}; // src/node.rs:36
    float win_scale = WIN_SCALE;
    uint8_t a = 0;
    float e = 0.0f;
    // NodeMutableState, src/node.rs:27-31
    std::optional<std::vector<float>> p;
    std::optional<std::vector<uint8_t>> v;
    std::optional<G> s;
    std::vector<size_t> children;

    Node() = default;
    explicit Node(float scale) : win_scale(scale) {}
    Node(const Node& o)
        : win_counter(o.win_counter.load()), win_scale(o.win_scale), a(o.a), e(o.e),
          p(o.p), v(o.v), s(o.s), children(o.children) {}
    Node& operator=(const Node& o) {
        win_counter.store(o.win_counter.load());
        win_scale = o.win_scale; a = o.a; e = o.e; p = o.p; v = o.v; s = o.s;
        children = o.children;
        return *this;
    }

    // src/node.rs:61-64
    float get_w() const {
        return (float)((int64_t)(win_counter.load() >> 32) - 0x7FFFFFFFll) / win_scale;
    }
    // src/node.rs:67-69
    uint16_t get_n() const { return (uint16_t)((win_counter.load() & 0x00000000FFFF0000ull) >> 16); }
    // src/node.rs:72-74
    uint16_t get_vloss() const { return (uint16_t)(win_counter.load() & 0xFFFFull); }
    // src/node.rs:51-58
    float compute_q() const {
        uint16_t n = get_n();
        if (n > 0) return (get_w() - (float)get_vloss()) / (float)n;
        return 0.0f;
    }
    // src/node.rs:77-80
    void visit() { win_counter.fetch_add(0x0000000000010001ull); }
    // exact inverse of visit(): an abandoned simulation of the lock-step schedule (repairs S11 / S12) leaves no trace
    void revert_visit() { win_counter.fetch_sub(0x0000000000010001ull); }
    // src/node.rs:83-92 (C4: a non-negative backup adds incr+1 to W; kept literally)
    void unvisit(float win_val) {
        uint32_t incr = (uint32_t)std::fabs(win_scale * win_val);
        uint64_t d = (win_val < 0.0f) ? ((uint64_t)incr << 32)
                                      : ((uint64_t)(0xFFFFFFFFu - incr) << 32);
        win_counter.fetch_sub(0x1ull | d);
    }
};

// src/node.rs:138-143
enum class NodeState { PlaceHolder, Locked, ExistsTrue, ExistsFalse };

// NodeStore: src/node.rs:129-375.  Fixed-capacity bump arena + lock flags +
// link slots + `seen` transposition map.
template <class G>
struct NodeStore {
    struct Slot {
        std::atomic<bool> lock{false};
        bool present = false;          // Option<NodeLink> is Some
        Node<G> node;
        std::optional<size_t> link;    // NodeLink.1
    };
    std::unique_ptr<Slot[]> buf;
    size_t cap = 0;
    std::atomic<size_t> len{0};
    std::unordered_map<G, size_t, typename G::Hasher> seen;

    // src/node.rs:146-154
    static std::unique_ptr<NodeStore> empty(size_t reserve_space) {
        auto ns = std::make_unique<NodeStore>();
        ns->buf.reset(new Slot[reserve_space]);
        ns->cap = reserve_space;
        return ns;
    }
    // src/node.rs:156-166
    static std::unique_ptr<NodeStore> make_new(size_t reserve_space) {
        return from_root(reserve_space, G::get_init_board());
    }
    // src/node.rs:168-177
    static std::unique_ptr<NodeStore> from_root(size_t reserve_space, const G& s) {
        auto ns = empty(reserve_space);
        size_t root_idx = ns->push(Node<G>(WIN_SCALE));
        ns->upgrade(root_idx, s);
        return ns;
    }
    // src/node.rs:179-193
    std::optional<size_t> resolve(size_t idx) const {
        size_t l = idx;
        size_t n = len.load();
        for (;;) {
            if (l >= n) return std::nullopt;
            const Slot& sl = buf[l];
            if (!sl.present) return std::nullopt;
            if (!sl.link) return l;
            l = *sl.link;
        }
    }
    // src/node.rs:195-201
    Node<G>* get(size_t idx) const {
        auto l = resolve(idx);
        return l ? &buf[*l].node : nullptr;
    }
    // un-resolved slot access (B3 repair: the edge action is the slot's own a)
    Node<G>* raw(size_t idx) const { return &buf[idx].node; }
    // src/node.rs:203-205
    std::optional<size_t> lookup_state_id(const G& s) const {
        auto it = seen.find(s);
        if (it == seen.end()) return std::nullopt;
        return it->second;
    }
    // src/node.rs:212-232
    bool set_policy(size_t idx, std::vector<float> policy) {
        auto l = resolve(idx);
        if (state(idx) != std::optional<NodeState>(NodeState::Locked)) return false;
        buf[*l].node.p = std::move(policy);
        return true;
    }
    // store a prior on a node that is not locked (S1 repair: root prior)
    void set_policy_unlocked(size_t idx, std::vector<float> policy) {
        buf[*resolve(idx)].node.p = std::move(policy);
    }
    // src/node.rs:234-244
    size_t push(const Node<G>& node) {
        size_t idx = len.fetch_add(1);
        if (idx >= cap) throw std::runtime_error("NodeStore: reserve exhausted"); // assert!, :237
        buf[idx].node = node;
        buf[idx].link.reset();
        buf[idx].present = true;
        return idx;
    }
    // src/node.rs:246-270
    std::optional<NodeState> state(size_t idx) const {
        if (idx >= len.load()) return std::nullopt;
        if (buf[idx].lock.load()) return NodeState::Locked;
        if (!buf[idx].present) return std::nullopt;
        if (!buf[idx].link) {
            return buf[idx].node.s ? NodeState::ExistsTrue : NodeState::PlaceHolder;
        }
        return NodeState::ExistsFalse;
    }
    // src/node.rs:272-326
    std::optional<bool> upgrade(size_t idx, const G& s) {
        if (idx >= len.load()) return std::nullopt;
        Slot& sl = buf[idx];
        assert(!sl.link);
        auto it = seen.find(s);
        if (it != seen.end()) {
            sl.link = it->second;               // :286
            unlock(idx);                        // :287
            return false;
        }
        Node<G>& nd = sl.node;
        nd.s = s;                               // :292
        float game_ended = s.get_game_ended(1); // :293
        nd.e = -game_ended;                     // :294 (C9)
        if (game_ended == 0.0f) {
            std::vector<uint8_t> valids = s.get_valid_moves(1); // :297
            std::vector<uint8_t> valid_actions;
            for (size_t i = 0; i < valids.size(); ++i)
                if (valids[i] != 0) valid_actions.push_back((uint8_t)i);
            nd.children.reserve(valid_actions.size());
            nd.v = valids;                      // :311
            for (uint8_t a : valid_actions) {   // :313-317, ascending action order (C7/C12)
                Node<G> child(WIN_SCALE);
                child.a = a;
                size_t ci = push(child);
                buf[idx].node.children.push_back(ci);
            }
        }
        seen.emplace(s, idx);                   // :320
        return true;
    }
    // src/node.rs:328-333
    bool lock(size_t idx) {
        bool expected = false;
        return buf[idx].lock.compare_exchange_strong(expected, true);
    }
    // src/node.rs:335-341 (debug_assert only: unlocking an unlocked slot is a no-op)
    void unlock(size_t idx) {
        bool expected = true;
        buf[idx].lock.compare_exchange_strong(expected, false);
    }
    // src/node.rs:343-370.  C6 operation order, C7 last-max ties, C8 filter,
    // B3 repair (edge action from the un-resolved slot).
    size_t best_child(size_t idx, int32_t cpuct, bool filter, bool b3_literal = false) const {
        const Node<G>* node = get(idx);
        uint16_t parent_n = node->get_n();
        bool have = false;
        size_t best = 0;
        float best_u = 0.0f;
        for (size_t child_idx : node->children) {
            const Node<G>* child = get(child_idx);
            uint8_t ea = b3_literal ? child->a : raw(child_idx)->a;
            float u = child->compute_q() +
                      (((float)cpuct * (*node->p)[ea]) * std::sqrt((float)parent_n + EPS)) /
                          (float)(uint16_t)(1 + child->get_n());
            if (filter && state(child_idx) == std::optional<NodeState>(NodeState::Locked)) continue;
            if (!have) { have = true; best = child_idx; best_u = u; continue; }
            // Iterator::max_by keeps the LATER element unless the earlier is
            // strictly Greater; NaN compares Equal (partial_cmp -> unwrap_or(Equal)).
            bool earlier_greater = best_u > u;
            if (!earlier_greater) { best = child_idx; best_u = u; }
        }
        if (!have) throw std::runtime_error("best_child: no children"); // .unwrap(), :367
        return best;
    }
    // true when best_child(idx, .., filter = true) would have nothing to choose from (every child Locked): the reference's
    // `max_by(..).unwrap()` panics there (:366-367); the lock-step schedule abandons the simulation instead (S11)
    bool all_children_locked(size_t idx) const {
        for (size_t child_idx : get(idx)->children)
            if (state(child_idx) != std::optional<NodeState>(NodeState::Locked)) return false;
        return true;
    }
    size_t size() const { return len.load(); }
};

// ---------------------------------------------------------------------------
// NNet: src/nnet.rs:35-45 (predict only; train is next-tier).
struct NNet {
    virtual ~NNet() = default;
    // boards [B, feat...] row-major f32 -> pi [B, A], v [B]
    virtual void predict(const float* boards, int B, int model_id, float* pi, float* v) = 0;
};

struct SearchStats {
    uint64_t sims = 0;        // search_iteration calls
    uint64_t expansions = 0;  // upgrade -> Some(true)
    uint64_t leaf_evals = 0;  // NNet::predict rows issued by the search (root priors included)
    uint64_t link_hits = 0;   // upgrade -> Some(false)
    uint64_t terminal_hits = 0;
    uint64_t depth_sum = 0;   // selection levels (best_child calls)
    uint64_t abandoned = 0;   // num_threads > 1: simulations abandoned (S11 / S12), counted in sims
};

// AsyncMcts: src/async_mcts.rs:14-115, :191-371.  num_threads == 1 (the only deterministic mode of the reference,
// examples/connect_four.rs:68) runs search_iteration as written; num_threads > 1 runs the reference's tree-parallel search
// (:191-217: num_threads OS threads sharing one NodeStore, virtual loss, Locked filter) as ONE LEGAL EXECUTION of it, a
// deterministic lock-step schedule (search_lockstep below).  The reference itself is racy for num_threads > 1, so nothing
// can pin this: PARITY UNPINNED.
template <class G>
struct AsyncMcts {
    size_t reserve_space;
    std::unique_ptr<NodeStore<G>> nodes;
    size_t num_sims, num_threads, max_depth, model_id;
    int32_t cpuct;
    NNet* net;
    Quirks quirks;
    SearchStats stats;
    size_t action_size;
    // one (state, pi, v) record per NN evaluation, for replay parity
    std::function<void(const G&, const float*, float)> on_eval;

    // src/async_mcts.rs:27-48
    AsyncMcts(size_t reserve, size_t sims, size_t threads, size_t maxd, size_t model,
              int32_t cp, NNet* n, size_t actions)
        : reserve_space(reserve), nodes(NodeStore<G>::make_new(reserve)), num_sims(sims),
          num_threads(threads), max_depth(maxd), model_id(model), cpuct(cp), net(n),
          action_size(actions) {}
    // src/async_mcts.rs:50-72
    AsyncMcts(const G& s, size_t reserve, size_t sims, size_t threads, size_t maxd,
              size_t model, int32_t cp, NNet* n, size_t actions)
        : reserve_space(reserve), nodes(NodeStore<G>::from_root(reserve, s)), num_sims(sims),
          num_threads(threads), max_depth(maxd), model_id(model), cpuct(cp), net(n),
          action_size(actions) {}

    // src/async_mcts.rs:300-345: featurise, predict, mask, renormalise (C10).
    std::pair<std::vector<float>, float> evaluate(const G& s, const std::vector<uint8_t>& valids) {
        std::vector<float> feat = s.to_features();
        std::vector<float> pi(action_size);
        float v = 0.0f;
        net->predict(feat.data(), 1, (int)model_id, pi.data(), &v);
        stats.leaf_evals++;
        if (on_eval) on_eval(s, pi.data(), v);
        for (size_t i = 0; i < action_size; ++i)
            if (valids[i] == 0) pi[i] = 0.0f;                       // :322-326
        float sum_ps = 0.0f;
        for (size_t i = 0; i < action_size; ++i) sum_ps = sum_ps + pi[i]; // ndarray sum, <8 elems: sequential
        if (sum_ps > 0.0f) {
            for (auto& x : pi) x = x / sum_ps;                      // :331
        } else {
            for (size_t i = 0; i < action_size; ++i) pi[i] = pi[i] + (float)valids[i]; // :340-342
            float s2 = 0.0f;
            for (size_t i = 0; i < action_size; ++i) s2 = s2 + pi[i];
            for (auto& x : pi) x = x / s2;                          // :344
        }
        return {pi, v};
    }

    // src/async_mcts.rs:74-115
    // Returns pi; fills counts/q (per action) when non-null.
    std::vector<float> get_action_prob(const G& s, float temp, uint64_t seed, uint64_t game_id,
                                       uint64_t ply, uint16_t* counts_out = nullptr,
                                       float* q_out = nullptr) {
        size_t root;
        auto found = nodes->lookup_state_id(s);                     // :81
        if (found) {
            root = *found;
        } else {
            // S10 (A11): unseen root state -> push + upgrade a fresh node, as from_root
            root = nodes->push(Node<G>(WIN_SCALE));
            nodes->upgrade(root, s);
            stats.expansions++;
        }
        {
            // S1 (A1): the root never receives a prior in the reference.
            Node<G>* rn = nodes->get(root);
            // a terminal root has no prior: the reference panics at :85 (p.unwrap())
            if (rn->e != 0.0f) throw std::runtime_error("get_action_prob: terminal root state");
            if (!rn->p) {
                auto pv = evaluate(*rn->s, *rn->v);
                nodes->set_policy_unlocked(root, std::move(pv.first));
            }
        }
        search(root);                                               // :82
        Node<G>* root_node = nodes->get(root);
        std::vector<uint16_t> counts(action_size, 0);               // :87
        std::vector<float> qs(action_size, 0.0f);
        for (size_t child_idx : root_node->children) {              // :88-94
            Node<G>* child = nodes->get(child_idx);
            uint8_t a = nodes->raw(child_idx)->a;                   // B3
            counts[a] = child->get_n();
            qs[a] = child->compute_q();
        }
        if (counts_out) for (size_t i = 0; i < action_size; ++i) counts_out[i] = counts[i];
        if (q_out) for (size_t i = 0; i < action_size; ++i) q_out[i] = qs[i];
        std::vector<float> probs(action_size, 0.0f);
        if (temp == 0.0f) {                                         // :97-107
            uint16_t max_val = 0;
            for (auto c : counts) if (c > max_val) max_val = c;
            std::vector<size_t> best;
            for (size_t i = 0; i < action_size; ++i) if (counts[i] == max_val) best.push_back(i);
            uint64_t r = rng_draw(seed, game_id, ply, RNG_TIEBREAK);
            size_t best_a = best[rng_choose(r, (uint32_t)best.size())];
            probs[best_a] = 1.0f;
            return probs;
        }
        // S6 (A7): probs[a] = counts[a]^(1/temp) / sum
        float inv_t = 1.0f / temp;
        std::vector<float> x(action_size);
        for (size_t i = 0; i < action_size; ++i)
            x[i] = (inv_t == 1.0f) ? (float)counts[i] : std::pow((float)counts[i], inv_t); // :109
        float sum = 0.0f;
        for (size_t i = 0; i < action_size; ++i) sum = sum + x[i];  // :110
        for (size_t i = 0; i < action_size; ++i) probs[i] = x[i] / sum;
        return probs;
    }

    // src/async_mcts.rs:191-217
    void search(size_t root_idx) {
        if (num_sims % num_threads != 0) throw std::runtime_error("num_sims % num_threads != 0");   // assert!, :192
        if (num_threads == 1 && !force_lockstep) {
            for (size_t i = 0; i < num_sims; ++i) search_iteration(root_idx);
        } else {
            search_lockstep(root_idx);
        }
    }
    bool force_lockstep = false;   // tests: run num_threads == 1 through search_lockstep (must equal search_iteration)

    // ---- num_threads > 1 as a deterministic lock-step schedule --------------------------------------------------------
    // The reference's threads each loop `while sim_id.fetch_add(1) < num_sims { search_iteration }` (:210-212) on one shared
    // NodeStore.  The schedule pinned here: num_sims / num_threads STEPS; in a step the threads run the select / expand part of
    // search_iteration (:226-309) one after the other in thread order -- so thread t sees the N + 1 and vloss + 1 of every
    // visit() (src/node.rs:77-80) the threads before it made, and the slots they still hold Locked --, then block on the
    // inference channel together (:311-315: the step's leaves are one batch), then finish (:317-353) and back up (:361-370)
    // in thread order.  C8 (:253-258, :275; src/node.rs:359-365): a Locked arg-max is retried with the Locked children
    // filtered out.  Two situations this schedule reaches have no legal continuation in the reference (it panics), and are
    // repaired by ABANDONING the simulation -- its visits are reverted exactly and it still counts toward num_sims:
    //   S11  every child of the node is Locked: `max_by` over an empty iterator, `.unwrap()` (src/node.rs:366-367)
    //   S12  a link (:293-299, or an Exists(false) child) leads to a node an earlier thread of the step holds Locked, i.e.
    //        expanded and still without its prior: `node.mu.p.as_ref().unwrap()` (src/node.rs:354)
    struct Pending {
        size_t cur = 0;
        std::vector<size_t> node_path;
        int kind = 0;               // 0 value known (terminal / depth), 1 leaf waits for the net, 2 abandoned
        float v = 0.0f;
    };
    Pending select_phase(size_t root_idx) {
        stats.sims++;
        Pending pd;
        pd.cur = root_idx;
        pd.node_path.reserve(64);
        size_t depth = 0;
        bool cur_visited = false;
        auto abandon = [&]() {
            for (size_t idx : pd.node_path) nodes->get(idx)->revert_visit();
            if (cur_visited) nodes->get(pd.cur)->revert_visit();
            pd.kind = 2;
            stats.abandoned++;
        };
        for (;;) {
            size_t cur = pd.cur;
            cur_visited = false;
            if (nodes->state(cur) == std::optional<NodeState>(NodeState::Locked)) { abandon(); return pd; }   // S12
            Node<G>* head = nodes->get(cur);
            head->visit();                                          // :251; S5
            cur_visited = true;
            if (depth > max_depth) { pd.v = head->s->eval_heuristic(); return pd; }   // :241-244 (B10)
            if (head->e != 0.0f) { pd.v = head->e; stats.terminal_hits++; return pd; } // :246-249
            size_t c = nodes->best_child(cur, cpuct, false);        // :255-258, first_iteration
            stats.depth_sum++;
            if (nodes->state(c) == std::optional<NodeState>(NodeState::Locked)) {     // `_ => continue`, :275
                if (nodes->all_children_locked(cur)) { abandon(); return pd; }        // S11
                c = nodes->best_child(cur, cpuct, true);            // C8: filter = !first_iteration
            }
            auto st = nodes->state(c);
            if (st == std::optional<NodeState>(NodeState::PlaceHolder)) {             // :261-268
                nodes->lock(c);                                     // succeeds: the schedule is sequential
                pd.node_path.push_back(cur);                        // S3
                size_t parent = cur;
                pd.cur = c;
                Node<G>* node_p = nodes->get(parent);
                uint8_t act = quirks.b1_parent_action ? node_p->a : nodes->raw(c)->a; // B1
                auto nx = node_p->s->get_next_state(1, act);        // :284
                G s2 = nx.first.get_canonical_form(nx.second);      // :287
                auto up = nodes->upgrade(c, s2);                    // :289
                if (!*up) {                                         // :293-299: link (upgrade unlocked the slot)
                    stats.link_hits++;
                    pd.cur = *nodes->resolve(c);
                    continue;
                }
                stats.expansions++;
                Node<G>* leaf = nodes->get(c);
                leaf->visit();                                      // :309
                if (leaf->e != 0.0f) {                              // S4
                    nodes->unlock(c);
                    pd.v = leaf->e;
                    return pd;
                }
                pd.kind = 1;                                        // stays Locked until its prior arrives (:311-351)
                return pd;
            } else {                                                // :269-274; S2
                pd.node_path.push_back(cur);
                pd.cur = *nodes->resolve(c);
                depth += 1;
            }
        }
    }
    void search_lockstep(size_t root_idx) {
        const size_t steps = num_sims / num_threads;
        for (size_t step = 0; step < steps; ++step) {
            std::vector<Pending> pend;
            pend.reserve(num_threads);
            for (size_t t = 0; t < num_threads; ++t) pend.push_back(select_phase(root_idx));
            for (auto& pd : pend) {                                 // the step's leaves: evaluated together, finished in thread order
                if (pd.kind != 1) continue;
                Node<G>* leaf = nodes->get(pd.cur);
                auto pv = evaluate(*leaf->s, *leaf->v);             // :303-345
                nodes->set_policy(pd.cur, std::move(pv.first));     // :348
                nodes->unlock(pd.cur);                              // :351
                pd.v = -pv.second;                                  // :353
            }
            for (auto& pd : pend) {                                 // backups, :361-370 (B2)
                if (pd.kind == 2) continue;
                float x = pd.v;
                nodes->get(pd.cur)->unvisit(x);
                while (!pd.node_path.empty()) {
                    size_t cur = pd.node_path.back();
                    pd.node_path.pop_back();
                    if (!quirks.b2_same_sign_backup) x = -x;
                    nodes->get(cur)->unvisit(x);
                }
            }
        }
    }

    // src/async_mcts.rs:219-371 with repairs S2-S5, B1, B2.
    void search_iteration(size_t root_idx) {
        stats.sims++;
        size_t cur = root_idx;
        std::vector<size_t> node_path;
        node_path.reserve(64);
        size_t depth = 0;
        float v;
        for (;;) {
            Node<G>* head = nodes->get(cur);
            head->visit();                                          // :251; S5 (A6): before the checks
            if (depth > max_depth) { v = head->s->eval_heuristic(); break; }  // :241-244 (B10)
            if (head->e != 0.0f) { v = head->e; stats.terminal_hits++; break; } // :246-249
            size_t c = nodes->best_child(cur, cpuct, false);        // :255-258 (single thread: filter never needed)
            stats.depth_sum++;
            auto st = nodes->state(c);
            if (st == std::optional<NodeState>(NodeState::PlaceHolder)) {     // :261-268
                nodes->lock(c);
                node_path.push_back(cur);                           // S3 (A3, A4)
                size_t parent = cur;
                cur = c;
                Node<G>* node_p = nodes->get(parent);
                uint8_t act = quirks.b1_parent_action ? node_p->a : nodes->raw(c)->a; // B1
                auto nx = node_p->s->get_next_state(1, act);        // :284
                G s2 = nx.first.get_canonical_form(nx.second);      // :287 (B5)
                auto up = nodes->upgrade(c, s2);                    // :289
                if (!*up) {                                         // :293-299 link
                    stats.link_hits++;
                    cur = *nodes->resolve(c);
                    continue;
                }
                stats.expansions++;
                Node<G>* leaf = nodes->get(c);
                leaf->visit();                                      // :309
                if (leaf->e != 0.0f) {                              // S4 (A5)
                    nodes->unlock(c);
                    v = leaf->e;
                    break;
                }
                auto pv = evaluate(*leaf->s, *leaf->v);             // :303-345
                nodes->set_policy(c, std::move(pv.first));          // :348
                nodes->unlock(c);                                   // :351
                v = -pv.second;                                     // :353
                break;
            } else {                                                // :269-274; S2 (A2): one level per iteration
                node_path.push_back(cur);
                cur = *nodes->resolve(c);
                depth += 1;
            }
        }
        // backup, src/async_mcts.rs:361-370; B2: alternate the sign toward the root
        float x = v;
        nodes->get(cur)->unvisit(x);
        while (!node_path.empty()) {
            cur = node_path.back();
            node_path.pop_back();
            if (!quirks.b2_same_sign_backup) x = -x;
            nodes->get(cur)->unvisit(x);
        }
    }
};

// ---------------------------------------------------------------------------
// arena: src/arena.rs:7-99
template <class G>
int8_t play_game(const std::function<uint8_t(const G&)>* player_actions, const std::optional<G>& board0) {
    int8_t cur_player = 1;
    G board = board0 ? *board0 : G::get_init_board();
    while (board.get_game_ended(cur_player) == 0.0f) {              // :18
        G canonical = board.get_canonical_form(cur_player);         // :25
        uint8_t action = player_actions[cur_player == 1 ? 0 : 1](canonical); // :27
        auto valids = canonical.get_valid_moves(1);                 // :29
        if (valids[action] == 0) throw std::runtime_error("arena: invalid action"); // :31-35
        auto nx = board.get_next_state(cur_player, action);         // :37
        board = nx.first;
        cur_player = nx.second;
    }
    return (int8_t)(cur_player * (int8_t)std::round(board.get_game_ended(cur_player))); // :51
}

struct GameResultCounts { uint64_t win = 0, loss = 0, draw = 0; };

// TrainingSample: src/nnet.rs:22-27
struct TrainingSample {
    std::vector<float> board;
    std::vector<float> pi;
    float v;
};

}  // namespace azo
