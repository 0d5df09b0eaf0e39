// az_tree.hip -- MCTS tree kernels for gfx950 (select / expand / leaf request / backup / root policy / self-play move),
// templates over the Game policy of az_game.h.  GROUP (8) lanes serve one game: lane j evaluates child j of the node
// being selected, so a node's child block (8 x 32-byte records, 256-byte aligned) is two coalesced 128-byte lines and
// the PUCT arg-max is an in-register fold over lane shuffles.  One simulation is in flight per tree (the reference's
// deterministic mode, num_sim_threads = 1), so every counter update is a plain read-modify-write by one lane:
// integer, order-free, bit-reproducible.
//
// Reference restated: src/async_mcts.rs:74-115, :219-371; src/node.rs:272-370;
// src/coach.rs:104-157; with the repairs of SURVEY.md section 0.2 (tagged S#/B#).
#include "az_tree.h"

namespace az {

// ---- lane-group primitives (GW = Game::GROUP lanes per tree) ------------------------------------------------------
template <int GW> AZ_D uint32_t gshfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, GW); }
template <int GW> AZ_D float gshflf(float v, int src) { return __shfl(v, src, GW); }
template <int GW> AZ_D uint64_t gshfl64(uint64_t v, int src) {
    return ((uint64_t)gshfl<GW>((uint32_t)(v >> 32), src) << 32) | gshfl<GW>((uint32_t)v, src);
}
template <int GW> AZ_D uint32_t gballot(bool p) {
    unsigned long long m = __ballot(p);
    int lane = threadIdx.x & 63;
    return (uint32_t)(m >> (lane & ~(GW - 1))) & ((1u << GW) - 1u);
}
template <int NA> AZ_D uint32_t nth_set_bit(uint32_t mask, uint32_t n) {
    uint32_t a = 0;
#pragma unroll
    for (int c = 0; c < NA; ++c) {
        bool set = (mask >> c) & 1u;
        if (set && n == 0) a = (uint32_t)c;
        if (set) --n;
    }
    return a;
}

// ---- node record access (layout: az_tree.h) ---------------------------------------------------------------------------
// The 16-byte record {ctr, prior, word}: word = a | nchild << 3 | ecode << 6 | has_prior << 8 | locked << 9 | kind << 10 | payload << 12,
// kind 0 = placeholder, 1 = link (payload = the canonical node's slot), 2 = expanded (payload = child block index).  NodeRec keeps the
// meta bits of az_common.h (EXPANDED = kind 2) so the search code reads as before; the packed state of an expanded node lives in t.key.
struct NodeRec {
    uint64_t ctr;
    uint32_t prior, meta, link, child_base;
};
AZ_HD uint32_t node_word(uint32_t meta, uint32_t link, uint32_t child_base) {
    const uint32_t kind = link != NONE ? 1u : ((meta & META_EXPANDED) ? 2u : 0u);
    const uint32_t payload = kind == 1u ? link : (kind == 2u ? child_base / (uint32_t)BLOCK_SLOTS : 0u);
    return (meta & 0x3Fu) | (((meta >> META_ECODE_SHIFT) & 0xFu) << 6) | (kind << 10) | (payload << 12);
}
AZ_D uint4* node_ptr(const TreeDev& t, size_t base, uint32_t slot) { return t.node + (base + slot); }
AZ_D NodeRec node_load(const uint4* p) {
    const uint4 a = p[0];
    const uint32_t w = a.w, kind = (w >> 10) & 3u, payload = w >> 12;
    NodeRec r;
    r.ctr = ((uint64_t)a.y << 32) | a.x;
    r.prior = a.z;
    r.meta = (w & 0x3Fu) | (((w >> 6) & 0xFu) << META_ECODE_SHIFT) | (kind == 2u ? META_EXPANDED : 0u);
    r.link = kind == 1u ? payload : NONE;
    r.child_base = kind == 2u ? payload * (uint32_t)BLOCK_SLOTS : 0u;
    return r;
}
AZ_D void node_store(const TreeDev& t, size_t base, uint32_t slot, uint64_t ctr, uint64_t key, uint32_t prior, uint32_t meta, uint32_t link,
                     uint32_t child_base) {
    t.node[base + slot] = make_uint4((uint32_t)ctr, (uint32_t)(ctr >> 32), prior, node_word(meta, link, child_base));
    if (meta & META_EXPANDED) t.key[base + slot] = key;           // only expanded nodes have a state (and only they are looked up by it)
}
AZ_D uint64_t node_ctr(const uint4* p) { return *(const unsigned long long*)p; }
AZ_D void node_set_ctr(uint4* p, uint64_t v) { *(unsigned long long*)p = v; }
AZ_D uint64_t node_key(const TreeDev& t, size_t base, uint32_t slot) { return t.key[base + slot]; }
AZ_D void node_set_prior(uint4* p, uint32_t bits) { ((uint32_t*)p)[2] = bits; }
// rewrite the word of a record whose fields the caller holds (meta changes, or a placeholder becoming a link)
AZ_D void node_set_word(uint4* p, uint32_t meta, uint32_t link, uint32_t child_base) { ((uint32_t*)p)[3] = node_word(meta, link, child_base); }

// `seen.get(&s)` (src/node.rs:282): GW-wide linear probe.  found = node slot or NONE;
// ins = first empty position (where `seen.insert` will go, src/node.rs:320).
template <class G>
AZ_D void hash_find(const TreeDev& t, int g, size_t base, typename G::State s, int sub, uint32_t* found, uint32_t* ins) {
    constexpr int GW = G::GROUP;
    const uint32_t mask = t.H - 1;
    const uint32_t h = G::hash(s) & mask;
    const uint64_t key = G::pack(s);
    const uint32_t* tab = t.hash + (size_t)g * t.H;
    for (uint32_t probe = 0; probe < t.H; probe += GW) {
        uint32_t pos = (h + probe + (uint32_t)sub) & mask;
        uint32_t idx = tab[pos];
        bool empty = idx == NONE;
        bool match = false;
        if (!empty) match = node_key(t, base, idx) == key;
        uint32_t em = gballot<GW>(empty), mm = gballot<GW>(match);
        int fe = em ? (__ffs((int)em) - 1) : GW;
        uint32_t before = mm & ((1u << fe) - 1u);
        if (before) {
            *found = gshfl<GW>(idx, __ffs((int)before) - 1);
            *ins = NONE;
            return;
        }
        if (em) {
            *found = NONE;
            *ins = (h + probe + (uint32_t)fe) & mask;
            return;
        }
    }
    *found = NONE;
    *ins = NONE;
}

// NodeStore::upgrade, `None` arm (src/node.rs:290-323): store s, e = -ended(s), push one placeholder per valid move in
// ascending action order into a fresh child block, insert into `seen`.  extra = nodes pushed besides the placeholders
// (1 when the upgraded node itself is new: a root).  Returns false when the arena is exhausted (assert!, src/node.rs:237).
template <class G>
AZ_D bool node_upgrade(const TreeDev& t, TreeHead& h, int g, size_t base, uint32_t slot, typename G::State s, uint32_t prior_bits,
                       uint32_t a, uint32_t ins_pos, uint64_t ctr_value, uint32_t cbase, uint32_t extra, int sub, uint32_t* ecode_out,
                       uint32_t lock_if_live = 0u /*META_LOCKED: the slot stays Locked until its prior is stored (src/node.rs:322, src/async_mcts.rs:351)*/) {
    uint32_t ec = G::ended_code(s);
    uint32_t vm = ec ? 0u : G::valid_mask(s);
    uint32_t nv = (uint32_t)__popc(vm);
    if (cbase + BLOCK_SLOTS > t.R || h.count + extra + nv > t.reserve_nodes || ins_pos == NONE) {
        if (sub == 0) atomicOr(&t.err[ins_pos == NONE ? ERR_HASH_FULL : ERR_CAPACITY], 1u);
        return false;
    }
    h.len = cbase + BLOCK_SLOTS;
    h.count += extra + nv;
    if (sub == 0) {
        node_store(t, base, slot, ctr_value, G::pack(s), prior_bits,
                   a | (nv << META_NCHILD_SHIFT) | META_EXPANDED | (ec << META_ECODE_SHIFT) | (ec == E_NONE ? lock_if_live : 0u), NONE, cbase);
        t.hash[(size_t)g * t.H + ins_pos] = slot;
    }
    if ((uint32_t)sub < nv)
        node_store(t, base, cbase + sub, CTR_INIT, 0ull, 0u, nth_set_bit<G::ACTIONS>(vm, (uint32_t)sub), NONE, 0u);
    *ecode_out = ec;
    return true;
}

// lane `sub`'s two inline node_path entries (j = sub and sub + 8)
struct PathRegs { uint32_t lo, hi; };
AZ_D PathRegs path_load(const TreeDev& t, int g, int sub) { return PathRegs{t.head[g].path16[sub], t.head[g].path16[8 + sub]}; }
AZ_D void path_store(const TreeDev& t, int g, int sub, const PathRegs& p) { t.head[g].path16[sub] = p.lo; t.head[g].path16[8 + sub] = p.hi; }

AZ_D TreeHead head_load(const TreeDev& t, int g) {
    const uint4* p = (const uint4*)(t.head + g);
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    TreeHead h;
    h.len = a.x; h.count = a.y; h.root = a.z; h.active = a.w;
    h.leaf = b.x; h.leaf_kind = b.y; h.leaf_val = __uint_as_float(b.z); h.src = b.w;
    h.path_len = c.x; h.log_len = c.y; h.stat[0] = c.z; h.stat[1] = c.w;
    h.stat[2] = d.x; h.stat[3] = d.y; h.stat[4] = d.z; h.stat[5] = d.w;
    return h;
}
AZ_D void head_store(const TreeDev& t, int g, const TreeHead& h) {
    uint4* p = (uint4*)(t.head + g);
    p[0] = make_uint4(h.len, h.count, h.root, h.active);
    p[1] = make_uint4(h.leaf, h.leaf_kind, __float_as_uint(h.leaf_val), h.src);
    p[2] = make_uint4(h.path_len, h.log_len, h.stat[0], h.stat[1]);
    p[3] = make_uint4(h.stat[2], h.stat[3], h.stat[4], h.stat[5]);
}

__global__ void k_init_heads(TreeDev t) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= t.G) return;
    TreeHead h{};
    h.active = 1;
    head_store(t, g, h);
}
__global__ void k_set_active(TreeDev t, uint32_t value) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < t.G) t.head[g].head.active = value;
}

// ---- NodeStore::new (src/node.rs:156-166): clear `seen`, push + upgrade the initial board ----
template <class G>
__global__ __launch_bounds__(256) void k_reset_trees(TreeDev t, const uint8_t* flags, uint8_t* clear_flags,
                                                     const ulonglong2* roots /*nullptr = initial board*/) {
    constexpr int GW = G::GROUP;
    int g = blockIdx.x;
    if (flags && !flags[g]) return;
    uint32_t* tab = t.hash + (size_t)g * t.H;
    for (uint32_t i = threadIdx.x; i < t.H; i += blockDim.x) tab[i] = NONE;
    __syncthreads();
    if (threadIdx.x < GW) {
        int sub = threadIdx.x;
        size_t base = (size_t)g * t.R;
        TreeHead h = head_load(t, g);
        h.len = 0;
        h.count = 0;
        uint32_t ec;
        const typename G::State rs = roots ? roots[g] : G::init();          // NodeStore::from_root, src/node.rs:168-177
        uint32_t ins = G::hash(rs) & (t.H - 1);
        // the root takes slot 0 of block 0, its children block 1
        node_upgrade<G>(t, h, g, base, 0u, rs, 0u, 0u, ins, CTR_INIT, BLOCK_SLOTS, 1u, sub, &ec);
        h.root = 0;
        h.log_len = 0;
        h.leaf_kind = LEAF_NONE;
        if (sub == 0) {
            head_store(t, g, h);
            if (clear_flags) clear_flags[g] = 0;
        }
    }
}

// ---- leaf request: inference batch assembly (src/async_mcts.rs:137-151 restated) + de-duplication ---------------------
// A tree whose leaf goes to the net (want) gets a SOURCE for its (pi, v):
//   1. the evaluation cache: the state's bucket is one 64-byte line of 8 keys, lane j checks way j;
//   2. the batch's election table: the first tree to CAS its key in wins a row, later ones point at the winner's slot;
//   3. a row of the batch: one atomicAdd per wave (8 trees), rows in lane order inside the wave.
// Without de-duplication every requesting tree goes straight to 3.  All lanes of the wave that are still alive call
// this together (the ballots are wave-wide); lanes of trees with nothing to evaluate pass want = false.
template <class G>
AZ_D uint32_t leaf_request(const EvalBatch& eb, const EvalCache& ec, bool want, typename G::State s, int sub) {
    constexpr int GW = G::GROUP;
    const int lane = (int)(threadIdx.x & 63);
    uint32_t src = 0;
    bool hit = false, dup = false, take = want;
    uint32_t tpos = 0;
    if (eb.dedup) {
        const unsigned long long key = (unsigned long long)G::pack(s);
        // the election table's first probe is fetched together with the cache's key line (one round trip, not two)
        const uint32_t pos0 = (uint32_t)(mix64(key) >> 24) & eb.tmask;
        unsigned long long cur0 = 0ull;
        if (want && sub == 0) cur0 = eb.tkey[pos0];
        if (ec.key) {
            const unsigned long long ck = key | ec.tag;
            const uint32_t bucket = (uint32_t)(mix64(ck) >> 20) & ec.bmask;
            const uint32_t ways = gballot<GW>(want && ec.key[(size_t)bucket * 8 + sub] == ck);
            if (want && ways) { hit = true; take = false; src = SRC_CACHE | (bucket * 8u + (uint32_t)__ffs((int)ways) - 1u); }
        }
        if (take && sub == 0) {
            const unsigned long long mine = key;                           // never 0 (Game::pack)
            uint32_t pos = pos0;
            unsigned long long cur = cur0;
            for (;; cur = eb.tkey[pos]) {
                if (cur == 0ull) {                                         // empty: try to take it
                    const unsigned long long prev = atomicCAS(&eb.tkey[pos], 0ull, mine);
                    if (prev == 0ull) break;                               // won the slot
                    cur = prev;                                            // somebody else took it first
                }
                if (cur == mine) { dup = true; break; }
                pos = (pos + 1u) & eb.tmask;
            }
            tpos = pos;
        }
        dup = gshfl<GW>(dup ? 1u : 0u, 0) != 0u;
        tpos = gshfl<GW>(tpos, 0);
        if (take && dup) { take = false; src = SRC_TABLE | tpos; }
    }
    // rows of the batch: one atomicAdd per wave
    const unsigned long long wm = __ballot(take && sub == 0);
    uint32_t base = 0;
    if (blockDim.x > 64) {
        // four waves per workgroup (k_backup_select): ONE atomicAdd on the batch's row counter per workgroup -- the counter is a single
        // hot address for the ~1000 waves of a launch.  Every wave of the workgroup gets here exactly once.
        __shared__ uint32_t s_cnt[4], s_base;
        const int w = (int)(threadIdx.x >> 6);
        if (lane == 0) s_cnt[w] = (uint32_t)__popcll(wm);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            s_base = total ? atomicAdd(eb.n, total) : 0u;
        }
        __syncthreads();
        base = s_base;
        for (int i = 0; i < w; ++i) base += s_cnt[i];
    } else if (wm) {
        const int leader = __ffsll((long long)wm) - 1;
        if (lane == leader) base = atomicAdd(eb.n, (uint32_t)__popcll(wm));
        base = (uint32_t)__shfl((int)base, leader, 64);
    }
    if (wm) {
        if (take) {
            const int lane0 = lane & ~(GW - 1);
            const uint32_t row = base + (uint32_t)__popcll(wm & ((1ull << lane0) - 1ull));
            if (sub == 0) {
                eb.state[row] = s;
                if (eb.dedup) eb.tuniq[tpos] = row;
            }
            src = row;
        }
    }
    if (ec.stat && eb.dedup) {
        const unsigned long long am = __ballot(want && sub == 0), hm = __ballot(hit && sub == 0), dm = __ballot(dup && want && sub == 0);
        if (am && lane == __ffsll((long long)am) - 1) {
            unsigned long long* st = ec.stat + (size_t)(blockIdx.x % DD_REPLICAS) * DD_STRIDE;
            atomicAdd(&st[DD_REQUESTED], (unsigned long long)__popcll(am));
            if (wm) atomicAdd(&st[DD_EXECUTED], (unsigned long long)__popcll(wm));
            if (hm) atomicAdd(&st[DD_CACHE_HITS], (unsigned long long)__popcll(hm));
            if (dm) atomicAdd(&st[DD_BATCH_DUPS], (unsigned long long)__popcll(dm));
        }
    }
    return src;
}

// ---- get_action_prob prologue: root lookup (src/async_mcts.rs:81) + S10 + S1 -------------
template <class G>
AZ_D typename G::State root_prepare_body(const TreeDev& t, TreeHead& h, const ulonglong2* root_states, int g, int sub) {
    const bool act = h.active != 0;
    size_t base = (size_t)g * t.R;
    typename G::State s = act ? root_states[g] : G::init();
    uint32_t kind = LEAF_NONE;
    if (act) {
        uint32_t found, ins;
        hash_find<G>(t, g, base, s, sub, &found, &ins);
        uint32_t root = found;
        uint32_t n_exp = 0;
        if (found == NONE) {
            // S10 (A11): unseen root -> push + upgrade a fresh node, as NodeStore::from_root (src/node.rs:168-177):
            // the root takes slot 0 of a new block, its children the next block
            uint32_t idx = h.len;
            uint32_t ec_;
            if (idx + BLOCK_SLOTS > t.R) {
                if (sub == 0) atomicOr(&t.err[ERR_CAPACITY], 1u);
            } else if (node_upgrade<G>(t, h, g, base, idx, s, 0u, 0u, ins, CTR_INIT, idx + BLOCK_SLOTS, 1u, sub, &ec_)) {
                root = idx;
                n_exp = 1;
            }
        }
        if (root != NONE) {
            uint32_t meta = (found == NONE) ? 0u : node_load(node_ptr(t, base, root)).meta;
            uint32_t ec_ = (found == NONE) ? G::ended_code(s) : ((meta >> META_ECODE_SHIFT) & 3u);
            if (ec_ != E_NONE) {
                // terminal root: the reference panics at root.mu.p.unwrap() (src/async_mcts.rs:85)
                if (sub == 0) atomicOr(&t.err[ERR_TERMINAL_ROOT], 1u);
            } else if (!(meta & META_HAS_PRIOR)) {
                kind = LEAF_ROOT;  // S1 (A1): evaluate the root once so best_child has a prior (not a simulation: nothing is backed up)
            }
        }
        h.root = (root == NONE) ? 0u : root;
        h.leaf = h.root;
        h.path_len = 0;
        if (root == NONE) h.active = 0;       // a failed root deactivates the tree for this search
        h.stat[ST_EXPANSIONS] += n_exp;
    }
    h.leaf_kind = kind;
    return s;
}

template <class G>
__global__ __launch_bounds__(64) void k_root_prepare(TreeDev t, EvalBatch eb, EvalCache ec, const ulonglong2* root_states) {
    constexpr int GW = G::GROUP;
    int tid = blockIdx.x * 64 + threadIdx.x;
    int g = tid / GW, sub = tid % GW;
    if (g >= t.G) return;
    TreeHead h = head_load(t, g);
    const typename G::State s = root_prepare_body<G>(t, h, root_states, g, sub);
    const uint32_t src = leaf_request<G>(eb, ec, h.leaf_kind == LEAF_ROOT, s, sub);
    if (h.leaf_kind == LEAF_ROOT) h.src = src;
    if (sub == 0) head_store(t, g, h);
}

// ---- search_iteration: select + expand (src/async_mcts.rs:226-299, Appendix A of SURVEY.md) ----
// MT (several simulations in flight per tree, src/async_mcts.rs:191-217 as a lock-step schedule): the thread sees the Locked slots
// earlier threads of the step hold.  C8 (src/async_mcts.rs:253-258, :275; src/node.rs:359-365): if the arg-max over all children is
// Locked, the retry excludes Locked children.  Two sites where the reference panics have no legal continuation and are repaired by
// ABANDONING the simulation (its visits are reverted exactly, it counts toward num_sims; *abandoned = 1):
//   S11  every child is Locked: `max_by` on an empty iterator -> `unwrap()` (src/node.rs:366-367)
//   S12  a link leads to a node that is Locked, i.e. expanded by an earlier thread of this step and still without its prior:
//        `p.as_ref().unwrap()` (src/node.rs:354)
template <class G, bool MT = false>
AZ_D typename G::State select_body(const TreeDev& t, TreeHead& h, PathRegs& pth, const SearchParams& sp, int g, int sub, uint32_t* path,
                                   uint32_t* abandoned = nullptr) {
    constexpr int GW = G::GROUP;
    constexpr int NA = G::ACTIONS;
    const bool act = h.active != 0;
    const size_t base = (size_t)g * t.R;
    bool give_up = false, cur_visited = false;
    uint32_t cur = h.root;
    uint32_t depth = 0, plen = 0, kind = LEAF_NONE;
    float val = 0.0f;
    uint32_t n_exp = 0, n_link = 0, n_term = 0, n_depth = 0;
    typename G::State leaf_s = G::init();
    // The record of an expanded child chosen at one level IS the parent of the next level and is already in the registers of
    // the lane that evaluated it: it is handed down by shuffles, so a level costs one dependent fetch (the child block), not
    // two.  Only the root and link targets (canonical nodes living elsewhere) are fetched by slot.
    NodeRec pr{};
    bool have_pr = false;
    while (act) {
        uint4* pp = node_ptr(t, base, cur);
        if (!have_pr) pr = node_load(pp);
        have_pr = false;
        if constexpr (MT) { if (pr.meta & META_LOCKED) { give_up = true; cur_visited = false; break; } }   // S12
        const uint64_t pc = pr.ctr + CTR_VISIT;                 // visit(), src/node.rs:77-80; S5: before the checks
        if (sub == 0) node_set_ctr(pp, pc);
        const uint32_t ecd = (pr.meta >> META_ECODE_SHIFT) & 3u;
        if (depth > sp.max_depth) { val = 0.0f; kind = LEAF_VALUE; break; }   // src/async_mcts.rs:241-244 (B10)
        if (ecd != E_NONE) { val = ecode_value(ecd); kind = LEAF_VALUE; ++n_term; break; }  // :246-249
        // best_child, src/node.rs:343-370
        const uint32_t nchild = (pr.meta >> META_NCHILD_SHIFT) & 7u, cb = pr.child_base;
        const float sq = puct_sqrt_parent(ctr_n(pc));
        NodeRec cr{0ull, 0u, 0u, NONE, 0u};
        float u = 0.0f;
        if ((uint32_t)sub < nchild) {
            // the child's record carries its own counter; only a link slot needs the second, dependent fetch of the
            // canonical node's counter (resolve(), src/node.rs:179-193)
            cr = node_load(node_ptr(t, base, cb + sub));
            uint64_t cc = cr.ctr;
            if (cr.link != NONE) cc = node_ctr(node_ptr(t, base, cr.link));
            u = puct(cc, __uint_as_float(cr.prior), sq, sp.cpuct_f);
        }
        uint32_t best = 0;
        float bu = gshflf<GW>(u, 0);
#pragma unroll
        for (int j = 1; j < NA; ++j) {                          // max_by: later element wins unless earlier is Greater (C7)
            float uj = gshflf<GW>(u, j);
            if ((uint32_t)j < nchild && !(bu > uj)) { best = (uint32_t)j; bu = uj; }
        }
        ++n_depth;
        if constexpr (MT) {
            const uint32_t lockm = gballot<GW>((uint32_t)sub < nchild && (cr.meta & META_LOCKED)) & ((1u << NA) - 1u);
            if ((lockm >> best) & 1u) {                         // C8: the winner is Locked -> retry without the Locked children
                bool have = false;
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    const float uj = gshflf<GW>(u, j);
                    if ((uint32_t)j >= nchild || ((lockm >> j) & 1u)) continue;
                    if (!have || !(bu > uj)) { best = (uint32_t)j; bu = uj; have = true; }
                }
                if (!have) { give_up = true; cur_visited = true; break; }       // S11
            }
        }
        const uint32_t clink = gshfl<GW>(cr.link, (int)best), cmeta = gshfl<GW>(cr.meta, (int)best), cprior = gshfl<GW>(cr.prior, (int)best);
        const uint32_t cslot = cb + best;
        if (plen >= (uint32_t)PATH_CAP) {
            if (sub == 0) atomicOr(&t.err[ERR_PATH], 1u);
            kind = LEAF_NONE;
            break;
        }
        if (plen < (uint32_t)PATH_INLINE) {                      // node_path.push, S3 / :270: lane plen & 7 keeps it
            if ((uint32_t)sub == (plen & 7u)) { if (plen < 8u) pth.lo = cur; else pth.hi = cur; }
        } else if (sub == 0) {
            path[plen] = cur;
        }
        ++plen;
        if (clink != NONE) { cur = clink; ++depth; continue; }  // Exists(false): follow the link (S2: one level per iteration)
        if (cmeta & META_EXPANDED) {                             // Exists(true)
            pr.ctr = gshfl64<GW>(cr.ctr, (int)best);
            pr.meta = cmeta;
            pr.child_base = gshfl<GW>(cr.child_base, (int)best);
            have_pr = true;
            cur = cslot;
            ++depth;
            continue;
        }
        // PlaceHolder (:261-268, S3): expand it.  B1: play the child's own action.
        // the parent's state: the one fetch of a simulation that goes to the key array (the levels above only read 16-byte records)
        const typename G::State s2 = G::play(G::unpack(node_key(t, base, cur)), (int)(cmeta & META_A_MASK));       // :284-287 (B5)
        uint32_t found, ins;
        hash_find<G>(t, g, base, s2, sub, &found, &ins);
        if (found != NONE) {                                    // upgrade -> Some(false): become a link (src/node.rs:285-289)
            if (sub == 0) node_set_word(node_ptr(t, base, cslot), cmeta, found, 0u);
            cur = found;
            ++n_link;
            continue;                                           // :297-298
        }
        uint32_t ec2;
        // the placeholder is visited right after the upgrade (:309): its counter becomes INIT + VISIT
        if (!node_upgrade<G>(t, h, g, base, cslot, s2, cprior, cmeta & META_A_MASK, ins, CTR_INIT + CTR_VISIT, h.len, 0u, sub, &ec2,
                             MT ? META_LOCKED : 0u)) {
            kind = LEAF_NONE;
            break;
        }
        ++n_exp;
        cur = cslot;
        if (ec2 != E_NONE) { val = ecode_value(ec2); kind = LEAF_VALUE; break; }   // S4 (A5)
        kind = LEAF_EVAL;                                       // :303-315: goes to the net
        leaf_s = s2;
        break;
    }
    if constexpr (MT) {
        if (give_up) {
            // revert this simulation's visits exactly (N - 1, vloss - 1): node_path[0 .. plen) and, if it was visited, cur.
            // Lane sub holds node_path entries sub and sub + 8; a line never repeats a node, so the lanes touch distinct counters.
            const uint32_t total = plen + (cur_visited ? 1u : 0u);
            for (uint32_t r = 0; r < total; r += GW) {
                const uint32_t i = r + (uint32_t)sub;
                if (i < total) {
                    const uint32_t node = i < plen ? (i < 8u ? pth.lo : (i < (uint32_t)PATH_INLINE ? pth.hi : path[i])) : cur;
                    uint4* np = node_ptr(t, base, node);
                    node_set_ctr(np, node_ctr(np) - CTR_VISIT);
                }
            }
            kind = LEAF_NONE;
            if (abandoned) *abandoned = 1u;
        }
    }
    if (act) {
        h.leaf = cur;
        h.leaf_val = val;
        h.path_len = plen;
        h.stat[ST_SIMS] += 1;
        h.stat[ST_EXPANSIONS] += n_exp;
        h.stat[ST_LINK_HITS] += n_link;
        h.stat[ST_TERMINAL_HITS] += n_term;
        h.stat[ST_DEPTH_SUM] += n_depth;
    }
    h.leaf_kind = kind;
    return leaf_s;
}

// `seen`-style sharing of evaluations across trees: a tree whose row was really evaluated publishes (pi, v) under its
// state's key.  Bucket = one 64-byte line of 8 keys.  The claim (a CAS of an empty way, starting at a way picked by the key's
// hash) is ISSUED as soon as the leaf's key is known and CHECKED at the end of the backup, so its round trip overlaps the
// prior / counter updates; a way taken by another key costs one more attempt on the next way (at most three).
struct CacheClaim {
    unsigned long long key, prev;
    uint32_t slot;
    bool open;
};
template <class G>
AZ_D CacheClaim cache_claim_begin(const EvalCache& ec, typename G::State s, int sub) {
    CacheClaim c{0ull, 0ull, 0u, false};
    if (G::stones(s) > ec.max_stones) return c;
    c.key = (unsigned long long)G::pack(s) | ec.tag;
    const unsigned long long hsh = mix64(c.key);
    c.slot = ((uint32_t)(hsh >> 20) & ec.bmask) * 8u + ((uint32_t)(hsh >> 50) & 7u);
    c.open = true;
    if (sub == 0) c.prev = atomicCAS(&ec.key[c.slot], 0ull, c.key);
    return c;
}
template <class G>
AZ_D void cache_claim_finish(const EvalCache& ec, CacheClaim c, float pv, int sub) {
    constexpr int GW = G::GROUP;
    bool inserted = false;
    if (c.open) {
        unsigned long long prev = gshfl64<GW>(c.prev, 0);
        for (int attempt = 0; prev != 0ull && prev != c.key && attempt < 2; ++attempt) {
            c.slot = (c.slot & ~7u) | ((c.slot + 1u) & 7u);
            if (sub == 0) prev = atomicCAS(&ec.key[c.slot], 0ull, c.key);
            prev = gshfl64<GW>(prev, 0);
        }
        if (prev == 0ull) {                               // claimed (prev == key: already published; else: bucket busy, not cached)
            ec.pv[(size_t)c.slot * 8 + sub] = pv;
            inserted = true;
        }
    }
    const unsigned long long im = __ballot(inserted && sub == 0);
    if (ec.stat && im && (int)(threadIdx.x & 63) == __ffsll((long long)im) - 1)
        atomicAdd(&ec.stat[(size_t)(blockIdx.x % DD_REPLICAS) * DD_STRIDE + DD_INSERTS], (unsigned long long)__popcll(im));
}

// ---- mask/renormalise/store the prior (src/async_mcts.rs:317-353) + backup (:361-370) ----
// INLINE_PV: the leaf's (pi, v) row is handed over in a register (pv_in: lane a < ACTIONS holds pi[a], lane ACTIONS holds v)
// instead of being read through TreeHead.src -- the fused search of the fixture nets.
template <class G, bool INLINE_PV = false>
AZ_D void backup_body(const TreeDev& t, TreeHead& h, const PathRegs& pth, const EvalBatch& eb, const EvalCache& ec, int g,
                      int sub, const uint32_t* path, float pv_in = 0.0f) {
    constexpr int GW = G::GROUP;
    constexpr int NA = G::ACTIONS;
    const uint32_t kind = h.leaf_kind;
    if (kind == LEAF_NONE) return;
    const size_t base = (size_t)g * t.R;
    const uint32_t leaf = h.leaf;
    float val;
    float pv = 0.0f;                                // lanes 0..NA-1: pi[sub], lane NA: v
    CacheClaim claim{0ull, 0ull, 0u, false};
    bool publish = false;
    const bool apply_only = kind == LEAF_ROOT;            // the root's own evaluation: store the prior, back nothing up
    if (kind == LEAF_EVAL || kind == LEAF_ROOT) {
        uint4* lp = node_ptr(t, base, leaf);
        const NodeRec lr = node_load(lp);
        const typename G::State s = G::unpack(node_key(t, base, leaf));
        const uint32_t src = h.src;
        if constexpr (INLINE_PV) {
            pv = pv_in;
        } else if (src & SRC_CACHE) {
            pv = ec.pv[(size_t)(src & SRC_INDEX) * 8 + sub];
        } else {
            const uint32_t row = (src & SRC_TABLE) ? eb.tuniq[src & SRC_INDEX] : src;
            pv = eb.pi[(size_t)row * 8 + sub];
            publish = !(src & SRC_TABLE) && eb.dedup && ec.key;
            if (publish) claim = cache_claim_begin<G>(ec, s, sub);
        }
        float p = sub < NA ? pv : 0.0f;
        const float v = gshflf<GW>(pv, NA);
        if (t.log_cap > 0) {
            uint32_t n = h.log_len;
            if (n < (uint32_t)t.log_cap) {
                size_t li = (size_t)(t.log_row ? t.log_row[g] : g) * t.log_cap + n;
                if (sub < NA) t.log_pi[li * NA + sub] = p;
                if (sub == 0) { t.log_v[li] = v; t.log_state[li] = s; }
            }
            h.log_len = n + 1;
        }
        const uint32_t vm = G::valid_mask(s);
        const bool valid = sub < NA && ((vm >> sub) & 1u);
        if (!valid) p = 0.0f;                                       // :322-326
        float sum = 0.0f;
#pragma unroll
        for (int a = 0; a < NA; ++a) sum = __fadd_rn(sum, gshflf<GW>(p, a));   // :328 (sequential, C10)
        if (sum > 0.0f) {
            p = __fdiv_rn(p, sum);                                  // :331
        } else {
            p = __fadd_rn(p, valid ? 1.0f : 0.0f);                  // :340-342
            float s2 = 0.0f;
#pragma unroll
            for (int a = 0; a < NA; ++a) s2 = __fadd_rn(s2, gshflf<GW>(p, a));
            p = __fdiv_rn(p, s2);                                   // :344
        }
        const uint32_t nchild = (lr.meta >> META_NCHILD_SHIFT) & 7u, cb = lr.child_base;
        const uint32_t myact = (uint32_t)sub < nchild ? nth_set_bit<NA>(vm, (uint32_t)sub) : 0u;
        const float pa = gshflf<GW>(p, (int)myact);
        if ((uint32_t)sub < nchild) node_set_prior(node_ptr(t, base, cb + sub), __float_as_uint(pa));   // set_policy, :348
        if (sub == 0) node_set_word(lp, (lr.meta | META_HAS_PRIOR) & ~META_LOCKED, lr.link, lr.child_base);      // set_policy + unlock, :348-351
        h.stat[ST_LEAF_EVALS] += 1;
        val = -v;                                                   // :353 (C9)
    } else {
        val = h.leaf_val;
    }
    if (!apply_only) {
        // unvisit() leaf -> root along node_path; B2: the sign alternates toward the root.
        // A Connect Four line never repeats a node, so the lanes update distinct counters.
        const uint32_t plen = h.path_len;
        for (uint32_t r = 0; r <= plen; r += GW) {                      // round r: lane sub takes step i = r + sub (group-uniform trips)
            const uint32_t i = r + (uint32_t)sub;
            const uint32_t j = i >= 1u && i <= plen ? plen - i : 0u;    // node_path index of step i >= 1
            const uint32_t lo = gshfl<GW>(pth.lo, (int)(j & 7u)), hi = gshfl<GW>(pth.hi, (int)(j & 7u));
            if (i <= plen) {
                uint32_t node = i == 0 ? leaf : (j < 8u ? lo : (j < (uint32_t)PATH_INLINE ? hi : path[j]));
                float x = (i & 1u) ? -val : val;
                uint4* np = node_ptr(t, base, node);
                node_set_ctr(np, node_ctr(np) - ctr_unvisit_delta(x));  // src/node.rs:83-92
            }
        }
    }
    if constexpr (!INLINE_PV) { if (publish) cache_claim_finish<G>(ec, claim, pv, sub); }
}

// ---- get_action_prob epilogue (src/async_mcts.rs:84-114): counts -> pi ----------------------
struct RootPolicy {
    float pi;        // this lane's action (sub < ACTIONS)
    uint32_t count;
    float q;
};
template <class G>
AZ_D RootPolicy root_policy(const TreeDev& t, uint32_t root, int g, int sub, float temp, uint64_t seed, uint64_t game_id, uint64_t ply) {
    constexpr int GW = G::GROUP;
    constexpr int NA = G::ACTIONS;
    const size_t base = (size_t)g * t.R;
    const NodeRec pr = node_load(node_ptr(t, base, root));
    const uint32_t nchild = (pr.meta >> META_NCHILD_SHIFT) & 7u, cb = pr.child_base;
    uint32_t ca = 0, cn = 0;
    float cq = 0.0f;
    if ((uint32_t)sub < nchild) {
        const NodeRec cr = node_load(node_ptr(t, base, cb + sub));
        const uint64_t cc = cr.link != NONE ? node_ctr(node_ptr(t, base, cr.link)) : cr.ctr;
        ca = cr.meta & META_A_MASK;                                 // B3: the slot's own action
        cn = ctr_n(cc);
        cq = ctr_q(cc);
    }
    RootPolicy out{0.0f, 0u, 0.0f};
#pragma unroll
    for (int j = 0; j < NA; ++j) {                                  // counts[a] = n, :88-94
        uint32_t aj = gshfl<GW>(ca, j), nj = gshfl<GW>(cn, j);
        float qj = gshflf<GW>(cq, j);
        if ((uint32_t)j < nchild && aj == (uint32_t)sub) { out.count = nj; out.q = qj; }
    }
    if (temp == 0.0f) {                                             // :97-107
        uint32_t mx = 0;
#pragma unroll
        for (int a = 0; a < NA; ++a) { uint32_t ca2 = gshfl<GW>(out.count, a); mx = ca2 > mx ? ca2 : mx; }
        uint32_t ties = gballot<GW>(sub < NA && out.count == mx) & ((1u << NA) - 1u);
        uint64_t r = rng_draw(seed, game_id, ply, RNG_TIEBREAK);
        uint32_t pick = nth_set_bit<NA>(ties, rng_choose(r, (uint32_t)__popc(ties)));
        out.pi = ((uint32_t)sub == pick) ? 1.0f : 0.0f;
    } else {                                                        // S6 (A7): counts^(1/temp) / sum
        float inv_t = __fdiv_rn(1.0f, temp);
        float x = (inv_t == 1.0f) ? (float)out.count : powf((float)out.count, inv_t);   // :109
        if (sub >= NA) x = 0.0f;
        float sum = 0.0f;
#pragma unroll
        for (int a = 0; a < NA; ++a) sum = __fadd_rn(sum, gshflf<GW>(x, a));           // :110
        out.pi = __fdiv_rn(x, sum);
    }
    return out;
}

// the batch a launch consumes gets its election table cleared (nothing reads the keys any more: the backups read tuniq)
AZ_D void clear_election_keys(const EvalBatch& eb) {
    if (!eb.dedup) return;
    const uint32_t total = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= eb.tmask; i += total) eb.tkey[i] = 0ull;
}
template <class G>
__global__ __launch_bounds__(64) void k_backup(TreeDev t, EvalBatch eb, EvalCache ec) {
    constexpr int GW = G::GROUP;
    const int tid = blockIdx.x * 64 + threadIdx.x;
    const int g = tid / GW, sub = tid % GW;
    if (tid == 0) { if (eb.max_n && *eb.n > *eb.max_n) *eb.max_n = *eb.n; *eb.n = 0; }   // the batch has been consumed (nothing in this kernel reads the count)
    clear_election_keys(eb);
    if (g >= t.G) return;
    TreeHead h = head_load(t, g);
    const PathRegs pth = path_load(t, g, sub);
    backup_body<G>(t, h, pth, eb, ec, g, sub, t.path + (size_t)g * PATH_CAP);
    h.leaf_kind = LEAF_NONE;
    if (sub == 0) head_store(t, g, h);
}

// backup of simulation i and select of simulation i+1 in one launch: both belong to the same 8 lanes of the same tree and
// nothing else touches that tree in between.  The leaf of i+1 goes into the OTHER eval batch (eb_next; its count was
// zeroed by the previous launch, this one zeroes eb_prev's), so the two ping-pong.
template <class G, bool STAMP = false>      // STAMP: diagnostic build, per-wave s_memtime stamps of the kernel's phases into dbg[wave][8]
__global__ __launch_bounds__(256) void k_backup_select(TreeDev t, EvalBatch eb_prev, EvalBatch eb_next, EvalCache ec,
                                                       SearchParams sp, unsigned long long* dbg) {
    constexpr int GW = G::GROUP;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;      // 64 or 256 threads (256 only when every wave of the grid holds trees)
    const int g = tid / GW, sub = tid % GW;
    unsigned long long ts[7] = {0, 0, 0, 0, 0, 0, 0};
#define AZ_TSTAMP(i_) if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ts[i_] = __builtin_amdgcn_s_memtime(); }
    AZ_TSTAMP(0);
    if (tid == 0) { if (eb_prev.max_n && *eb_prev.n > *eb_prev.max_n) *eb_prev.max_n = *eb_prev.n; *eb_prev.n = 0; }
    clear_election_keys(eb_prev);
    if (g >= t.G) return;
    TreeHead h = head_load(t, g);
    PathRegs pth = path_load(t, g, sub);
    AZ_TSTAMP(1);
    backup_body<G>(t, h, pth, eb_prev, ec, g, sub, t.path + (size_t)g * PATH_CAP);
    AZ_TSTAMP(2);
    // the counters this tree's other lanes just wrote are read by the selection below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    AZ_TSTAMP(3);
    const typename G::State leaf_s = select_body<G>(t, h, pth, sp, g, sub, t.path + (size_t)g * PATH_CAP);
    AZ_TSTAMP(4);
    const uint32_t src = leaf_request<G>(eb_next, ec, h.leaf_kind == LEAF_EVAL, leaf_s, sub);
    AZ_TSTAMP(5);
    if (h.leaf_kind == LEAF_EVAL) h.src = src;
    if (sub == 0) head_store(t, g, h);
    path_store(t, g, sub, pth);
    AZ_TSTAMP(6);
#undef AZ_TSTAMP
    if constexpr (STAMP) {
        if (dbg && (threadIdx.x & 63) == 0 && h.active && h.leaf_kind == LEAF_EVAL) {      // a wave whose first tree searched and asked for a row
            unsigned long long* o = dbg + (size_t)(tid >> 6) * 8;
            for (int i = 0; i < 6; ++i) o[i] = ts[i + 1] - ts[i];
            o[6] = ts[6] - ts[0];
        }
    }
}

AZ_D void group_memory_sync() {
    // what one lane of the group stored (node records, node_path, counters) is read by its other lanes next
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- several simulations in flight per tree: ONE lock-step step (src/async_mcts.rs:191-217 as a deterministic schedule) ----------
// The T threads of a tree are served by the tree's 8 lanes one after the other.  Backups of the previous step's leaves in thread
// order (mask/renorm/store the prior, unlock, unvisit along the thread's own node_path), then T selections in thread order: thread
// tt sees the N + 1 / vloss + 1 of the threads before it (visit(), src/node.rs:77-80; Q = (W - vloss) / N, :51-58) and their Locked
// leaves (C8).  Each thread's leaf request goes into the step's batch, so a step evaluates up to T rows per tree together.
// The per-thread part of the search state lives in t.thr[g*T + tt]; t.head[g] keeps what the threads share.
struct ThreadRegs { uint32_t leaf, leaf_kind; float leaf_val; uint32_t src, path_len; };
AZ_D void thread_to_head(TreeHead& h, const ThreadRegs& r) { h.leaf = r.leaf; h.leaf_kind = r.leaf_kind; h.leaf_val = r.leaf_val; h.src = r.src; h.path_len = r.path_len; }
AZ_D ThreadRegs head_to_thread(const TreeHead& h) { return ThreadRegs{h.leaf, h.leaf_kind, h.leaf_val, h.src, h.path_len}; }
template <class G>
__global__ __launch_bounds__(64) void k_step_mt(TreeDev t, EvalBatch eb_prev, EvalBatch eb_next, EvalCache ec, SearchParams sp, int first, int last) {
    constexpr int GW = G::GROUP;
    const int tid = blockIdx.x * 64 + threadIdx.x;
    const int g = tid / GW, sub = tid % GW;
    if (tid == 0) { if (eb_prev.max_n && *eb_prev.n > *eb_prev.max_n) *eb_prev.max_n = *eb_prev.n; *eb_prev.n = 0; }
    clear_election_keys(eb_prev);
    if (g >= t.G) return;
    TreeHead h = head_load(t, g);
    const ThreadRegs root_req = head_to_thread(h);          // first: k_root_prepare left the root's evaluation request in the head (S1)
    const int T = t.T;
    for (int tt = 0; tt < T; ++tt) {                        // backups in thread order
        TreeLine* tl = t.thr + (size_t)g * T + tt;
        ThreadRegs r{0u, LEAF_NONE, 0.0f, 0u, 0u};
        PathRegs pth{0u, 0u};
        if (first) { if (tt == 0) r = root_req; }
        else {
            const uint4 a = ((const uint4*)tl)[1];
            r = ThreadRegs{a.x, a.y, __uint_as_float(a.z), a.w, tl->head.path_len};
            pth = PathRegs{tl->path16[sub], tl->path16[8 + sub]};
        }
        thread_to_head(h, r);
        backup_body<G>(t, h, pth, eb_prev, ec, g, sub, t.path + ((size_t)g * T + tt) * PATH_CAP);
    }
    h.leaf_kind = LEAF_NONE;
    for (int tt = 0; tt < T; ++tt) {                        // selections in thread order
        group_memory_sync();                                // counters, priors, locks and links written so far are read next
        TreeLine* tl = t.thr + (size_t)g * T + tt;
        ThreadRegs r{0u, LEAF_NONE, 0.0f, 0u, 0u};
        PathRegs pth{0u, 0u};
        uint32_t abandoned = 0u;
        bool want = false;
        typename G::State leaf_s = G::init();
        if (!last) {
            leaf_s = select_body<G, true>(t, h, pth, sp, g, sub, t.path + ((size_t)g * T + tt) * PATH_CAP, &abandoned);
            want = h.leaf_kind == LEAF_EVAL;
        }
        const uint32_t src = leaf_request<G>(eb_next, ec, want, leaf_s, sub);      // every wave calls it T times (wave-wide ballots inside)
        if (want) h.src = src;
        r = head_to_thread(h);
        if (last) r.leaf_kind = LEAF_NONE;
        if (sub == 0) {
            ((uint4*)tl)[1] = make_uint4(r.leaf, r.leaf_kind, __float_as_uint(r.leaf_val), r.src);
            tl->head.path_len = r.path_len;
            if (abandoned) tl->head.stat[0] += 1u;
        }
        tl->path16[sub] = pth.lo;
        tl->path16[8 + sub] = pth.hi;
        h.leaf_kind = LEAF_NONE;
    }
    if (sub == 0) head_store(t, g, h);
}

// ---- the whole get_action_prob search in ONE launch, for nets that are a pure function of the state on the device ----
// (DumbConnectFourNnet, examples/connect_four.rs:12-43, and the hash fixture): root prepare, then num_sims x {select,
// evaluate in registers, backup} per tree with no kernel boundary and no leaf batch -- the trees never wait for each other.
// Same operations per tree in the same order as the launch-per-step path: bit-identical.
template <class G>
AZ_D float fixture_row(typename G::State s, int kind, uint64_t salt, int sub) {
    float pi[G::ACTIONS], v;
    if (kind == 0) {
#pragma unroll
        for (int a = 0; a < G::ACTIONS; ++a) pi[a] = __fdiv_rn(1.0f, (float)G::ACTIONS);   // examples/connect_four.rs:34-41 (S9)
        v = 1.0f;
    } else {
        hashnet_eval(s.x, s.y, salt, pi, &v);
    }
    float out = v;
#pragma unroll
    for (int a = 0; a < G::ACTIONS; ++a) out = sub == a ? pi[a] : out;
    return out;
}
template <class G>
__global__ __launch_bounds__(64) void k_search_fixture(TreeDev t, const ulonglong2* root_states, SearchParams sp, int num_sims, int kind,
                                                       uint64_t salt) {
    constexpr int GW = G::GROUP;
    const int tid = blockIdx.x * 64 + threadIdx.x;
    const int g = tid / GW, sub = tid % GW;
    if (g >= t.G) return;
    const EvalBatch no_eb{};
    const EvalCache no_ec{};
    TreeHead h = head_load(t, g);
    PathRegs pth{0u, 0u};
    typename G::State ls = root_prepare_body<G>(t, h, root_states, g, sub);
    for (int i = 0; i <= num_sims; ++i) {
        group_memory_sync();
        // backup of the previous leaf (i == 0: the root's priors only, S1), then the next selection
        const float pv = (h.leaf_kind == LEAF_EVAL || h.leaf_kind == LEAF_ROOT) ? fixture_row<G>(ls, kind, salt, sub) : 0.0f;
        backup_body<G, true>(t, h, pth, no_eb, no_ec, g, sub, t.path + (size_t)g * PATH_CAP, pv);
        if (i == num_sims) break;
        group_memory_sync();
        ls = select_body<G>(t, h, pth, sp, g, sub, t.path + (size_t)g * PATH_CAP);
    }
    h.leaf_kind = LEAF_NONE;
    if (sub == 0) head_store(t, g, h);
}

template <class G>
__global__ __launch_bounds__(64) void k_root_policy(TreeDev t, float temp, uint64_t seed, uint64_t first_game_id,
                                                    float* pi, uint16_t* counts, float* q) {
    constexpr int GW = G::GROUP;
    constexpr int NA = G::ACTIONS;
    int tid = blockIdx.x * 64 + threadIdx.x;
    int g = tid / GW, sub = tid % GW;
    if (g >= t.G) return;
    const TreeHead h = head_load(t, g);
    if (!h.active) return;
    const typename G::State s = G::unpack(node_key(t, (size_t)g * t.R, h.root));
    RootPolicy rp = root_policy<G>(t, h.root, g, sub, temp, seed, first_game_id + (uint64_t)g, (uint64_t)G::stones(s));
    if (sub < NA) {
        pi[(size_t)g * NA + sub] = rp.pi;
        if (counts) counts[(size_t)g * NA + sub] = (uint16_t)rp.count;
        if (q) q[(size_t)g * NA + sub] = rp.q;
    }
}

// sums the per-tree counters into totals (one atomicAdd per counter per wave) and clears them
__global__ __launch_bounds__(256) void k_harvest(TreeDev t, unsigned long long* totals, uint32_t* node_counts) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t st[ST_TOTALS] = {0, 0, 0, 0, 0, 0, 0};
    if (g < t.G) {
        TreeHead* hp = &t.head[g].head;
#pragma unroll
        for (int k = 0; k < ST_COUNT; ++k) { st[k] = hp->stat[k]; hp->stat[k] = 0; }
        if (node_counts) node_counts[g] = hp->count;
        if (t.thr)
            for (int tt = 0; tt < t.T; ++tt) { TreeHead* tp = &t.thr[(size_t)g * t.T + tt].head; st[ST_ABANDONED] += tp->stat[0]; tp->stat[0] = 0; }
    }
#pragma unroll
    for (int k = 0; k < ST_TOTALS; ++k) {
        unsigned long long v = st[k];                 // 64 lanes x < 2^32 each: the sum fits 38 bits
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, 64), lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, 64);
            v += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 0 && v) atomicAdd(&totals[k], v);
    }
}

__global__ __launch_bounds__(256) void k_call_readback(unsigned long long* totals, unsigned long long* dd_stat, const uint32_t* err,
                                                       CallReadback* out) {
    __shared__ unsigned long long dd[DD_COUNT];
    const int tid = threadIdx.x;
    if (tid < DD_COUNT) dd[tid] = 0;
    __syncthreads();
    if (dd_stat && tid < DD_REPLICAS) {
#pragma unroll
        for (int i = 0; i < DD_COUNT; ++i) {
            const unsigned long long v = dd_stat[tid * DD_STRIDE + i];
            if (v) { atomicAdd(&dd[i], v); dd_stat[tid * DD_STRIDE + i] = 0; }
        }
    }
    __syncthreads();
    if (tid < ST_TOTALS) { out->totals[tid] = totals[tid]; totals[tid] = 0; }
    if (tid < DD_COUNT) out->dd[tid] = dd[tid];
    if (tid < ERR_COUNT) out->err[tid] = err[tid];
}

// ---- Coach::execute_episode, one ply of one slot (src/coach.rs:118-156) -------------------
// Returns 0: the game goes on (gd.state / player / ply advanced); 1: the episode ended and the slot got the next one (its tree must be
// rebuilt: need_reset); 2: the episode ended and the slot stays idle (h.active = 0).
template <class G>
AZ_D int selfplay_move_body(const TreeDev& t, TreeHead& h, const GamesDev& gd, const SelfplayMoveParams& mp, int g, int sub) {
    constexpr int GW = G::GROUP;
    constexpr int NA = G::ACTIONS;
    const int gi = gd.gid[g];
    const int ply = gd.ply[g];
    const int8_t player = gd.player[g];
    const typename G::State s = gd.state[g];
    const uint64_t game_id = mp.first_game_id + (uint64_t)gi;
    const float temp = (ply + 1 < mp.temp_threshold) ? 1.0f : 0.0f;         // :122-126 (episode_step = ply + 1)
    RootPolicy rp = root_policy<G>(t, h.root, g, sub, temp, mp.seed, game_id, (uint64_t)ply);   // :128
    const size_t so = (size_t)gi * G::MAX_PLIES + ply;
    if (sub < NA) gd.smp_pi[so * NA + sub] = rp.pi;                         // :130-135 (symmetries regenerated at emit)
    // choose_weighted, :137-138
    float w[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) w[a] = gshflf<GW>(rp.pi, a);
    float total = 0.0f;
#pragma unroll
    for (int a = 0; a < NA; ++a) total = __fadd_rn(total, w[a]);
    const uint64_t r = rng_draw(mp.seed, game_id, (uint64_t)ply, RNG_MOVE);
    const float uu = (float)(uint32_t)(r >> 40) * (1.0f / 16777216.0f);
    const float target = __fmul_rn(uu, total);
    float acc = 0.0f;
    int action = -1, last = -1;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        if (w[a] > 0.0f) {
            acc = __fadd_rn(acc, w[a]);
            last = a;
            if (action < 0 && target < acc) action = a;
        }
    }
    if (action < 0) action = last;
    if (action < 0) action = 0;
    const typename G::State s2 = G::play(s, action);                        // :140-142
    const uint32_t ec = G::ended_code(s2);                                  // r = get_game_ended(cur_player), :144
    int status = 0;
    if (sub == 0) {
        gd.smp_state[so] = s;
        gd.smp_player[so] = player;
        gd.moves[so] = (uint8_t)action;
        if (ec != E_NONE) {
            // canonical ended(s') = -e: -1 when the side to move has lost, DRAW_EPS on a full board
            gd.g_result[gi] = -ecode_value(ec);
            gd.g_final_player[gi] = (int8_t)-player;
            gd.g_len[gi] = ply + 1;
            if (gd.g_log_len) gd.g_log_len[gi] = (int32_t)h.log_len;
            if (gi >= mp.done_lo && gi < mp.done_hi) atomicAdd(&gd.counters[1], 1u);
            int next = -1;
            if (mp.refill) {
                uint32_t nx = atomicAdd(&gd.counters[0], 1u);
                if (nx < (uint32_t)gd.n_games) next = (int)nx;
            }
            gd.gid[g] = next;
            if (next >= 0) {
                gd.state[g] = G::init();
                gd.player[g] = 1;
                gd.ply[g] = 0;
                gd.need_reset[g] = 1;
                status = 1;
            } else {
                atomicSub(&gd.counters[2], 1u);
                status = 2;
            }
        } else {
            gd.state[g] = s2;
            gd.player[g] = (int8_t)-player;
            gd.ply[g] = ply + 1;
        }
    }
    status = (int)gshfl<GW>((uint32_t)status, 0);
    if (status == 2) h.active = 0;
    return status;
}
template <class G>
__global__ __launch_bounds__(64) void k_selfplay_move(TreeDev t, GamesDev gd, SelfplayMoveParams mp) {
    constexpr int GW = G::GROUP;
    int tid = blockIdx.x * 64 + threadIdx.x;
    int g = tid / GW, sub = tid % GW;
    if (g >= t.G) return;
    const int gi = gd.gid[g];
    TreeHead h = head_load(t, g);
    if (gi < 0 || !h.active) return;
    if (selfplay_move_body<G>(t, h, gd, mp, g, sub) == 2 && sub == 0) t.head[g].head.active = 0;
}

// ---- FREE-RUNNING self-play: every slot on its own timeline ("selfplay_async") -------------------------------------------------------
// The lock-step driver runs one simulation per tree per forward, and every tree waits for the move boundary of all: of 8192 trees only
// ~40 % put a row into a step's batch (the others hit the evaluation cache, a terminal node or a transposition), so the net runs on
// batches of ~2900 rows.  Here a slot advances by itself: one launch takes a tree through backup -> [move, next root] -> select ...
// until its next leaf needs the net (bounded by max_iters), and several launches share one leaf batch: trees whose leaf was answered by
// the cache go on in the NEXT launch (the payload of an entry is only read one launch after its key was seen, as everywhere), trees
// that took a row (or point at another tree's row) stay parked until the forward.  Moves, episode ends and slot refills happen inside the
// kernel; a refilled slot's tree is rebuilt by k_reset_trees between two launches.  Each game still depends on (seed, game id) and the
// net's rows alone -- the schedule decides when a row is evaluated, never what it is.
//   gd.sims[g]   simulations of the current move done so far; -1 = the move's root is not prepared yet
//   first        the first launch behind a forward: eb_prev holds that forward's rows (parked trees back up; its table is cleared)
template <class G>
__global__ __launch_bounds__(256) void k_async_step(TreeDev t, GamesDev gd, EvalBatch eb_prev, EvalBatch eb_next, EvalCache ec, SearchParams sp,
                                                    SelfplayMoveParams mp, int num_sims, int first, int max_iters) {
    constexpr int GW = G::GROUP;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;      // whole workgroups only (the launcher pads nothing: leaf_request synchronises)
    const int g = tid / GW, sub = tid % GW;
    if (first) {
        if (tid == 0) { if (eb_prev.max_n && *eb_prev.n > *eb_prev.max_n) *eb_prev.max_n = *eb_prev.n; *eb_prev.n = 0; }
        clear_election_keys(eb_prev);
    }
    if (g >= t.G) return;
    TreeHead h = head_load(t, g);
    PathRegs pth = path_load(t, g, sub);
    uint32_t* path = t.path + (size_t)g * PATH_CAP;
    int sims = gd.sims[g];
    const bool alive = gd.gid[g] >= 0 && h.active != 0 && !gd.need_reset[g];
    // a tree whose leaf waits for a row of the batch being filled stays parked; a leaf answered by the cache (or a value) can go on
    const bool parked = h.leaf_kind != LEAF_NONE && h.leaf_kind != LEAF_VALUE && !(h.src & SRC_CACHE) && !first;
    typename G::State leaf_s = G::init();
    bool want = false;
    if (alive && !parked) {
        for (int it = 0; it < max_iters; ++it) {
            if (h.leaf_kind != LEAF_NONE) {                                    // the pending leaf: store its prior, back its value up
                const bool was_sim = h.leaf_kind != LEAF_ROOT;
                group_memory_sync();
                backup_body<G>(t, h, pth, eb_prev, ec, g, sub, path);
                h.leaf_kind = LEAF_NONE;
                if (was_sim) ++sims;
                continue;
            }
            group_memory_sync();
            if (sims < 0) {                                                    // get_action_prob's prologue for the slot's position (S10, S1)
                leaf_s = root_prepare_body<G>(t, h, gd.state, g, sub);
                sims = 0;
                if (!h.active) break;                                          // a failed root (error flag set): the call ends with an error
                if (h.leaf_kind == LEAF_ROOT) { want = true; break; }
                continue;
            }
            if (sims >= num_sims) {                                            // the move (src/coach.rs:128-156), then the next position's root
                const int st = selfplay_move_body<G>(t, h, gd, mp, g, sub);
                sims = -1;
                if (st != 0) break;                                            // episode over: idle, or wait for k_reset_trees
                continue;
            }
            leaf_s = select_body<G>(t, h, pth, sp, g, sub, path);
            if (h.leaf_kind == LEAF_EVAL) { want = true; break; }
            if (h.leaf_kind == LEAF_NONE) ++sims;                              // an error cut the simulation short (flag set): it still counts
        }
    }
    const uint32_t src = leaf_request<G>(eb_next, ec, want, leaf_s, sub);
    if (want) h.src = src;
    if (sub == 0) { head_store(t, g, h); gd.sims[g] = sims; }
    path_store(t, g, sub, pth);
}

// ---- training tuples (TrainingSample, src/nnet.rs:22-27; z per B4, src/coach.rs:146-154) -----
template <class G>
__global__ __launch_bounds__(256) void k_emit_samples(GamesDev gd, const int64_t* offsets, int symmetries,
                                                      ulonglong2* out_states, float* out_boards, float* out_pis,
                                                      float* out_zs) {
    constexpr int NA = G::ACTIONS;
    constexpr int NF = G::FEATURES;
    static_assert(NF > NA + 1, "the item loop below spreads pi / z / state stores over the first feature indices");
    const int gi = blockIdx.x;
    const int len = gd.g_len[gi];
    const int nsym = symmetries ? 2 : 1;
    const float r = gd.g_result[gi];
    const int8_t fin = gd.g_final_player[gi];
    for (int item = threadIdx.x; item < len * nsym * NF; item += blockDim.x) {
        const int f = item % NF, rest = item / NF;
        const int sym = rest % nsym, ply = rest / nsym;
        const size_t so = (size_t)gi * G::MAX_PLIES + ply;
        const int64_t o = (offsets[gi] + ply) * nsym + sym;
        typename G::State s = gd.smp_state[so];
        if (sym) s = G::mirror(s);                                             // get_symmetries, connect_four_game.rs:205-211
        if (out_boards) out_boards[o * NF + f] = G::feature(s, f);
        if (f < NA) out_pis[o * NA + f] = gd.smp_pi[so * NA + (sym ? G::mirror_action(f) : f)];
        if (f == NA) out_zs[o] = __fmul_rn(r, gd.smp_player[so] == fin ? 1.0f : -1.0f);   // B4
        if (f == NA + 1 && out_states) out_states[o] = s;
    }
}

// ---- arena::play_game, one ply for every running game (src/arena.rs:18-41) ------------------------
// Which model moves: seat 0 moves when cur_player == +1; games g < half seat (new, old), the rest (old, new).
__global__ void k_arena_sync(TreeDev tn, TreeDev to, ArenaDev ad) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ad.G) return;
    const bool alive = ad.alive[g] != 0;
    const int first_model = ad.first + g < ad.half ? 0 : 1;            // 0 = new, 1 = old (global game index)
    const int mover = ad.player[g] == 1 ? first_model : 1 - first_model;
    tn.head[g].head.active = (alive && mover == 0) ? 1u : 0u;
    to.head[g].head.active = (alive && mover == 1) ? 1u : 0u;
}

// The searching tree's owner plays argmax(get_action_prob(s, temp = 0)) (src/coach.rs:356-372).
template <class G>
__global__ __launch_bounds__(64) void k_arena_move(TreeDev t, ArenaDev ad, uint64_t seed) {
    constexpr int GW = G::GROUP;
    constexpr int NA = G::ACTIONS;
    int tid = blockIdx.x * 64 + threadIdx.x;
    int g = tid / GW, sub = tid % GW;
    if (g >= t.G) return;
    const TreeHead h = head_load(t, g);
    if (!h.active) return;
    const typename G::State s = ad.state[g];
    const int8_t player = ad.player[g];
    RootPolicy rp = root_policy<G>(t, h.root, g, sub, 0.0f, seed, (uint64_t)(ad.first + g), (uint64_t)G::stones(s));
    // argmax with max_by (last max) over the one-hot pi = the index of the 1
    const uint32_t hot = gballot<GW>(sub < NA && rp.pi == 1.0f) & ((1u << NA) - 1u);
    const int action = hot ? (31 - __clz((int)hot)) : 0;
    const bool valid = (G::valid_mask(s) >> action) & 1u;             // src/arena.rs:29-35
    const typename G::State s2 = G::play(s, action);
    const uint32_t ec = G::ended_code(s2);
    if (sub == 0) {
        if (valid && hot) {                           // the game's move record
            const int32_t k = ad.len[g];
            if (k < G::MAX_PLIES) ad.moves[(size_t)g * G::MAX_PLIES + k] = (uint8_t)action;
            ad.len[g] = k + 1;
        }
        if (!valid || !hot) {
            atomicOr(&ad.counters[1], 1u);
            ad.alive[g] = 0;
            atomicSub(&ad.counters[0], 1u);
        } else if (ec != E_NONE) {
            // cur_player' = -player; result = cur_player' * round(get_game_ended(cur_player')) (src/arena.rs:51):
            // ended = -1 -> the player who just moved won; DRAW_EPS rounds to 0
            ad.results[g] = (ec == E_PLUS1) ? player : (ec == E_MINUS1 ? (int8_t)-player : (int8_t)0);
            ad.alive[g] = 0;
            atomicSub(&ad.counters[0], 1u);
        } else {
            ad.state[g] = s2;
            ad.player[g] = (int8_t)-player;
        }
    }
}

__global__ void k_sync_active(TreeDev t, GamesDev gd) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < t.G) t.head[g].head.active = gd.gid[g] >= 0 ? 1u : 0u;
}

// ---- launchers: one instantiation of every kernel per Game policy, chosen by TreeDev.game ------------------------------
template <class G> constexpr bool game_ok() {
    return G::GROUP == BLOCK_SLOTS && G::GROUP > G::ACTIONS && (G::GROUP & (G::GROUP - 1)) == 0 && sizeof(typename G::State) == 16 &&
           sizeof(typename G::Packed) == 8;
}
static_assert(game_ok<ConnectFour>() && game_ok<ConnectThree>(),
              "one lane per child plus one, power of two, one child block per group; 16-byte states, 8-byte node-resident identity");
#define AZ_FOR_GAME(game_, ...)                                    \
    do {                                                           \
        if ((game_) == 1) { using TG = ConnectThree; __VA_ARGS__; } \
        else { using TG = ConnectFour; __VA_ARGS__; }               \
    } while (0)
static inline int group_blocks(int G) { return (G * BLOCK_SLOTS + 63) / 64; }
#ifdef AZ_DIAG
// diagnostic library only: the PUCT term of best_child (src/node.rs:352-356) for n (child counter, prior bits, parent N) triples, as the
// selection kernels compute it -- lets a test hold every operation of it to the oracle's IEEE arithmetic, bit for bit
__global__ void k_diag_puct(const unsigned long long* ctr, const uint32_t* prior, const uint32_t* parent_n, float cpuct, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = puct(ctr[i], __uint_as_float(prior[i]), puct_sqrt_parent(parent_n[i]), cpuct);
}
extern "C" int az_diag_puct(const unsigned long long* ctr, const uint32_t* prior, const uint32_t* parent_n, int cpuct, float* out, int n) {
    unsigned long long* d_c = nullptr; uint32_t *d_p = nullptr, *d_n = nullptr; float* d_o = nullptr;
    bool ok = hipMalloc((void**)&d_c, (size_t)n * 8) == hipSuccess && hipMalloc((void**)&d_p, (size_t)n * 4) == hipSuccess &&
              hipMalloc((void**)&d_n, (size_t)n * 4) == hipSuccess && hipMalloc((void**)&d_o, (size_t)n * 4) == hipSuccess;
    ok = ok && hipMemcpy(d_c, ctr, (size_t)n * 8, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d_p, prior, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_n, parent_n, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_diag_puct, dim3((n + 255) / 256), dim3(256), 0, nullptr, d_c, d_p, d_n, (float)cpuct, d_o, n);
        ok = hipMemcpy(out, d_o, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d_c); (void)hipFree(d_p); (void)hipFree(d_n); (void)hipFree(d_o);
    return ok ? 0 : -1;
}
// diagnostic library only: per-wave phase stamps of k_backup_select (tools/tree_probe.py); process-wide by design (a probe, not a product path)
constexpr int TREE_DBG_WAVES = 4096;
static unsigned long long* g_tree_dbg = nullptr;
bool tree_set_stamps(int on) {
    if (on && !g_tree_dbg) return hipMalloc((void**)&g_tree_dbg, (size_t)TREE_DBG_WAVES * 8 * sizeof(unsigned long long)) == hipSuccess &&
                                  hipMemset(g_tree_dbg, 0, (size_t)TREE_DBG_WAVES * 8 * sizeof(unsigned long long)) == hipSuccess;
    if (!on && g_tree_dbg) { (void)hipFree(g_tree_dbg); g_tree_dbg = nullptr; }
    return true;
}
bool tree_read_stamps(unsigned long long* out /*[TREE_DBG_WAVES * 8]*/) {
    return g_tree_dbg && hipMemcpy(out, g_tree_dbg, (size_t)TREE_DBG_WAVES * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess;
}
#endif

void launch_init_heads(const TreeDev& t, hipStream_t s) {
    hipLaunchKernelGGL(k_init_heads, dim3((t.G + 255) / 256), dim3(256), 0, s, t);
}
void launch_set_active(const TreeDev& t, uint32_t value, hipStream_t s) {
    hipLaunchKernelGGL(k_set_active, dim3((t.G + 255) / 256), dim3(256), 0, s, t, value);
}
void launch_reset_trees(const TreeDev& t, const uint8_t* flags, hipStream_t s, const ulonglong2* roots) {
    AZ_FOR_GAME(t.game, hipLaunchKernelGGL(k_reset_trees<TG>, dim3(t.G), dim3(256), 0, s, t, flags, const_cast<uint8_t*>(flags), roots));
}
void launch_root_prepare(const TreeDev& t, const EvalBatch& eb, const EvalCache& ec, const ulonglong2* root_states, hipStream_t s) {
    AZ_FOR_GAME(t.game, hipLaunchKernelGGL(k_root_prepare<TG>, dim3(group_blocks(t.G)), dim3(64), 0, s, t, eb, ec, root_states));
}
void launch_backup(const TreeDev& t, const EvalBatch& eb, const EvalCache& ec, hipStream_t s) {
    AZ_FOR_GAME(t.game, hipLaunchKernelGGL(k_backup<TG>, dim3(group_blocks(t.G)), dim3(64), 0, s, t, eb, ec));
}
void launch_backup_select(const TreeDev& t, const EvalBatch& eb_prev, const EvalBatch& eb_next, const EvalCache& ec, SearchParams sp,
                          hipStream_t s) {
    const bool four = t.block4 && (t.G * 8) % 256 == 0;     // whole 4-wave workgroups: one row-counter atomic per workgroup (leaf_request)
    const dim3 grid(four ? (unsigned)(t.G * 8 / 256) : group_blocks(t.G)), block(four ? 256 : 64);
#ifdef AZ_DIAG
    if (g_tree_dbg && t.G * 8 / 64 <= TREE_DBG_WAVES) {
        AZ_FOR_GAME(t.game, hipLaunchKernelGGL((k_backup_select<TG, true>), grid, block, 0, s, t, eb_prev, eb_next, ec, sp, g_tree_dbg));
        return;
    }
#endif
    AZ_FOR_GAME(t.game, hipLaunchKernelGGL((k_backup_select<TG, false>), grid, block, 0, s, t, eb_prev, eb_next, ec, sp, nullptr));
}
void launch_step_mt(const TreeDev& t, const EvalBatch& eb_prev, const EvalBatch& eb_next, const EvalCache& ec, SearchParams sp,
                    int first, int last, hipStream_t s) {
    // one wave per workgroup: leaf_request is called T times per launch and its one-atomic-per-workgroup path keeps state in LDS
    AZ_FOR_GAME(t.game, hipLaunchKernelGGL(k_step_mt<TG>, dim3(group_blocks(t.G)), dim3(64), 0, s, t, eb_prev, eb_next, ec, sp, first, last));
}
void launch_search_fixture(const TreeDev& t, const ulonglong2* root_states, SearchParams sp, int num_sims, int kind, uint64_t salt,
                           hipStream_t s) {
    AZ_FOR_GAME(t.game, hipLaunchKernelGGL(k_search_fixture<TG>, dim3(group_blocks(t.G)), dim3(64), 0, s, t, root_states, sp, num_sims, kind, salt));
}
void launch_root_policy(const TreeDev& t, float temp, uint64_t seed, uint64_t first_game_id, float* pi,
                        uint16_t* counts, float* q, hipStream_t s) {
    AZ_FOR_GAME(t.game, hipLaunchKernelGGL(k_root_policy<TG>, dim3(group_blocks(t.G)), dim3(64), 0, s, t, temp, seed, first_game_id, pi, counts, q));
}
void launch_harvest(const TreeDev& t, unsigned long long* totals, uint32_t* node_counts, hipStream_t s) {
    hipLaunchKernelGGL(k_harvest, dim3((t.G + 255) / 256), dim3(256), 0, s, t, totals, node_counts);
}
void launch_call_readback(unsigned long long* totals, unsigned long long* dd_stat, const uint32_t* err, CallReadback* out, hipStream_t s) {
    static_assert(DD_REPLICAS <= 256, "one thread per replica");
    hipLaunchKernelGGL(k_call_readback, dim3(1), dim3(256), 0, s, totals, dd_stat, err, out);
}
void launch_selfplay_move(const TreeDev& t, const GamesDev& gd, SelfplayMoveParams mp, hipStream_t s) {
    AZ_FOR_GAME(t.game, hipLaunchKernelGGL(k_selfplay_move<TG>, dim3(group_blocks(t.G)), dim3(64), 0, s, t, gd, mp));
}
void launch_async_step(const TreeDev& t, const GamesDev& gd, const EvalBatch& eb_prev, const EvalBatch& eb_next, const EvalCache& ec, SearchParams sp,
                       SelfplayMoveParams mp, int num_sims, int first, int max_iters, hipStream_t s) {
    const bool four = t.block4 && (t.G * 8) % 256 == 0;
    const dim3 grid(four ? (unsigned)(t.G * 8 / 256) : group_blocks(t.G)), block(four ? 256 : 64);
    AZ_FOR_GAME(t.game, hipLaunchKernelGGL(k_async_step<TG>, grid, block, 0, s, t, gd, eb_prev, eb_next, ec, sp, mp, num_sims, first, max_iters));
}
void launch_selfplay_sync_active(const TreeDev& t, const GamesDev& gd, hipStream_t s) {
    hipLaunchKernelGGL(k_sync_active, dim3((t.G + 255) / 256), dim3(256), 0, s, t, gd);
}
void launch_arena_sync(const TreeDev& t_new, const TreeDev& t_old, const ArenaDev& ad, hipStream_t s) {
    hipLaunchKernelGGL(k_arena_sync, dim3((ad.G + 255) / 256), dim3(256), 0, s, t_new, t_old, ad);
}
void launch_arena_move(const TreeDev& t, const ArenaDev& ad, uint64_t seed, hipStream_t s) {
    AZ_FOR_GAME(t.game, hipLaunchKernelGGL(k_arena_move<TG>, dim3(group_blocks(t.G)), dim3(64), 0, s, t, ad, seed));
}
// episodes of [lo, hi) that have already finished (g_len is set when an episode ends): a session's chunk starts its count from here
__global__ void k_count_done(const int32_t* __restrict__ g_len, int lo, int hi, uint32_t* __restrict__ counter) {
    __shared__ uint32_t part[256];
    uint32_t n = 0;
    for (int i = lo + (int)threadIdx.x; i < hi; i += 256) n += g_len[i] > 0 ? 1u : 0u;
    part[threadIdx.x] = n;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *counter = part[0];
}
void launch_count_done(const int32_t* g_len, int lo, int hi, uint32_t* counter, hipStream_t s) {
    hipLaunchKernelGGL(k_count_done, dim3(1), dim3(256), 0, s, g_len, lo, hi, counter);
}
void launch_emit_samples(const GamesDev& gd, const int64_t* offsets, int symmetries, ulonglong2* out_states,
                         float* out_boards, float* out_pis, float* out_zs, hipStream_t s) {
    // features, mirror and plies are the board's, not the rules': both games share ConnectFour's
    hipLaunchKernelGGL(k_emit_samples<ConnectFour>, dim3(gd.n_games), dim3(256), 0, s, gd, offsets, symmetries, out_states,
                       out_boards, out_pis, out_zs);
}

}  // namespace az
