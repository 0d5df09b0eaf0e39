// az_tree.hip -- MCTS tree kernels for gfx950 (select / expand / compact / backup /
// root policy / self-play move).  8 lanes serve one game: lane j evaluates child j
// of the node being selected, so a node's <=7 contiguous child records are one
// coalesced 16-byte-per-lane access and the PUCT arg-max is a 7-step in-register
// fold over lane shuffles.  One simulation is in flight per tree (the reference's
// deterministic mode, num_sim_threads = 1), so every counter update is a plain
// read-modify-write by one lane: integer, order-free, bit-reproducible.
//
// Reference restated: src/async_mcts.rs:74-115, :219-371; src/node.rs:272-370;
// src/coach.rs:104-157; with the repairs of SURVEY.md section 0.2 (tagged S#/B#).
#include "az_tree.h"

namespace az {

// ---- 8-lane group primitives -----------------------------------------------------
AZ_D uint32_t gshfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, LANES); }
AZ_D float gshflf(float v, int src) { return __shfl(v, src, LANES); }
AZ_D uint32_t gballot(bool p) {
    unsigned long long m = __ballot(p);
    int lane = threadIdx.x & 63;
    return (uint32_t)(m >> (lane & ~(LANES - 1))) & 0xFFu;
}
AZ_D uint32_t nth_set_bit(uint32_t mask, uint32_t n) {
    uint32_t a = 0;
#pragma unroll
    for (int c = 0; c < ACTIONS; ++c) {
        bool set = (mask >> c) & 1u;
        if (set && n == 0) a = (uint32_t)c;
        if (set) --n;
    }
    return a;
}

// `seen.get(&s)` (src/node.rs:282): 8-wide linear probe.  found = node index or NONE;
// ins = first empty position (where `seen.insert` will go, src/node.rs:320).
AZ_D void hash_find(const TreeDev& t, int g, size_t base, uint64_t m, uint64_t th, int sub, uint32_t* found,
                    uint32_t* ins) {
    const uint32_t mask = t.H - 1;
    const uint32_t h = c4_hash(m, th) & mask;
    const uint32_t* tab = t.hash + (size_t)g * t.H;
    for (uint32_t probe = 0; probe < t.H; probe += LANES) {
        uint32_t pos = (h + probe + (uint32_t)sub) & mask;
        uint32_t idx = tab[pos];
        bool empty = idx == NONE;
        bool match = false;
        if (!empty) {
            ulonglong2 s = t.state[base + idx];
            match = s.x == m && s.y == th;
        }
        uint32_t em = gballot(empty), mm = gballot(match);
        int fe = em ? (__ffs((int)em) - 1) : LANES;
        uint32_t before = mm & ((1u << fe) - 1u);
        if (before) {
            *found = gshfl(idx, __ffs((int)before) - 1);
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

// NodeStore::upgrade, `None` arm (src/node.rs:290-323): store s, e = -ended(s), push one
// placeholder per valid move in ascending action order, insert into `seen`.
// cbase = current bump pointer (children go to [cbase, cbase+nv)).  Returns false when the
// arena is exhausted (assert!, src/node.rs:237).
AZ_D bool node_upgrade(const TreeDev& t, int g, size_t base, uint32_t slot, uint64_t m, uint64_t th,
                       uint32_t prior_bits, uint32_t a, uint32_t ins_pos, uint64_t ctr_value, uint32_t cbase, int sub,
                       uint32_t* ecode_out) {
    uint32_t ec = c4_ecode(m, th);
    uint32_t vm = ec ? 0u : c4_valid_mask(m, th);
    uint32_t nv = (uint32_t)__popc(vm);
    if (cbase + nv > t.R || ins_pos == NONE) {
        if (sub == 0) atomicOr(&t.err[ins_pos == NONE ? ERR_HASH_FULL : ERR_CAPACITY], 1u);
        return false;
    }
    if (sub == 0) {
        t.len[g] = cbase + nv;
        t.state[base + slot] = make_ulonglong2(m, th);
        t.rec[base + slot] =
            make_uint4(NONE, prior_bits, a | (nv << META_NCHILD_SHIFT) | META_EXPANDED | (ec << META_ECODE_SHIFT), cbase);
        t.ctr[base + slot] = ctr_value;
        t.hash[(size_t)g * t.H + ins_pos] = slot;
    }
    if ((uint32_t)sub < nv) {
        t.rec[base + cbase + sub] = make_uint4(NONE, 0u, nth_set_bit(vm, (uint32_t)sub), 0u);
        t.ctr[base + cbase + sub] = CTR_INIT;
    }
    *ecode_out = ec;
    return true;
}

// ---- NodeStore::new (src/node.rs:156-166): clear `seen`, push + upgrade the initial board ----
__global__ __launch_bounds__(256) void k_reset_trees(TreeDev t, const uint8_t* flags, uint8_t* clear_flags,
                                                     const ulonglong2* roots /*nullptr = initial board*/) {
    int g = blockIdx.x;
    if (flags && !flags[g]) return;
    uint32_t* tab = t.hash + (size_t)g * t.H;
    for (uint32_t i = threadIdx.x; i < t.H; i += blockDim.x) tab[i] = NONE;
    __syncthreads();
    if (threadIdx.x < LANES) {
        int sub = threadIdx.x;
        size_t base = (size_t)g * t.R;
        uint32_t ec;
        const ulonglong2 rs = roots ? roots[g] : make_ulonglong2(0ull, 0ull);     // NodeStore::from_root, src/node.rs:168-177
        uint32_t ins = c4_hash(rs.x, rs.y) & (t.H - 1);
        node_upgrade(t, g, base, 0u, rs.x, rs.y, 0u, 0u, ins, CTR_INIT, 1u, sub, &ec);
        if (sub == 0) {
            t.root[g] = 0;
            t.log_len[g] = 0;
            if (clear_flags) clear_flags[g] = 0;
        }
    }
}

// ---- inference batch assembly (src/async_mcts.rs:137-151 restated) -------------------------------------------------
// A tree whose leaf goes to the net takes the next row of the eval batch: one atomicAdd per wave (8 trees), rows in lane
// order inside the wave.  The row ORDER of a batch therefore depends on which wave's atomic lands first; the results do
// not, because a row's (pi, v) is independent of its position and of the batch's composition (tests/test_net_gpu.py,
// test_net_rows_are_batch_independent) and every tree finds its own row through slot_of.  k_backup resets the counter.
// (A separate single-block compaction kernel with rows in tree order cost 15.6 us per simulation step: 25 % of the
// tree-only time, 1.3 % with the conv net.)
AZ_D void batch_append(const TreeDev& t, const EvalBatch& eb, int g, bool want, uint64_t m_, uint64_t t_) {
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long mask = __ballot(want);
    if (!mask) return;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(eb.n, (uint32_t)__popcll(mask));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (want) {
        const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        t.slot_of[g] = (int32_t)slot;
        eb.tree[slot] = (uint32_t)g;
        eb.state[slot] = make_ulonglong2(m_, t_);
    }
}

// ---- get_action_prob prologue: root lookup (src/async_mcts.rs:81) + S10 + S1 -------------
__global__ __launch_bounds__(64) void k_root_prepare(TreeDev t, EvalBatch eb, const ulonglong2* root_states) {
    int tid = blockIdx.x * 64 + threadIdx.x;
    int g = tid >> 3, sub = tid & 7;
    if (g >= t.G) return;
    if (!t.active[g]) {
        if (sub == 0) t.leaf_kind[g] = LEAF_NONE;
        return;
    }
    size_t base = (size_t)g * t.R;
    ulonglong2 s = root_states[g];
    uint32_t found, ins;
    hash_find(t, g, base, s.x, s.y, sub, &found, &ins);
    uint32_t kind = LEAF_NONE;
    uint32_t root = found;
    uint32_t n_exp = 0;
    if (found == NONE) {
        // S10 (A11): unseen root -> push + upgrade a fresh node, as NodeStore::from_root (src/node.rs:168-177)
        uint32_t idx = t.len[g];
        uint32_t ec;
        if (idx + 1 > t.R) {
            if (sub == 0) atomicOr(&t.err[ERR_CAPACITY], 1u);
        } else if (node_upgrade(t, g, base, idx, s.x, s.y, 0u, 0u, ins, CTR_INIT, idx + 1, sub, &ec)) {
            root = idx;
            n_exp = 1;
        }
    }
    if (root != NONE) {
        uint32_t meta = (found == NONE) ? 0u : t.rec[base + root].z;
        uint32_t ec = (found == NONE) ? c4_ecode(s.x, s.y) : ((meta >> META_ECODE_SHIFT) & 3u);
        if (ec != E_NONE) {
            // terminal root: the reference panics at root.mu.p.unwrap() (src/async_mcts.rs:85)
            if (sub == 0) atomicOr(&t.err[ERR_TERMINAL_ROOT], 1u);
        } else if (!(meta & META_HAS_PRIOR)) {
            kind = LEAF_EVAL;  // S1 (A1): evaluate the root once so best_child has a prior
        }
    }
    if (sub == 0) {
        t.root[g] = (root == NONE) ? 0u : root;
        t.leaf[g] = (root == NONE) ? 0u : root;
        t.leaf_kind[g] = kind;
        t.path_len[g] = 0;
        if (root == NONE || kind == LEAF_NONE) {
            // nothing to evaluate; a failed root also deactivates the tree for this search
            if (root == NONE) t.active[g] = 0;
        }
        t.stat[(size_t)g * ST_COUNT + ST_EXPANSIONS] += n_exp;
    }
    batch_append(t, eb, g, sub == 0 && kind == LEAF_EVAL, s.x, s.y);
}

// ---- search_iteration: select + expand (src/async_mcts.rs:226-299, Appendix A of SURVEY.md) ----
AZ_D void select_body(const TreeDev& t, const EvalBatch& eb, const SearchParams& sp, int g, int sub) {
    if (!t.active[g]) {
        if (sub == 0) t.leaf_kind[g] = LEAF_NONE;
        return;
    }
    const size_t base = (size_t)g * t.R;
    uint32_t* path = t.path + (size_t)g * PATH_CAP;
    const uint32_t len_g = t.len[g];
    uint32_t cur = t.root[g];
    uint32_t depth = 0, plen = 0, kind = LEAF_NONE;
    float val = 0.0f;
    uint32_t n_exp = 0, n_link = 0, n_term = 0, n_depth = 0;
    uint64_t leaf_m = 0, leaf_t = 0;
    for (;;) {
        uint4 pr = t.rec[base + cur];
        uint64_t pc = t.ctr[base + cur] + CTR_VISIT;            // visit(), src/node.rs:77-80; S5: before the checks
        if (sub == 0) t.ctr[base + cur] = pc;
        uint32_t ec = (pr.z >> META_ECODE_SHIFT) & 3u;
        if (depth > sp.max_depth) { val = 0.0f; kind = LEAF_VALUE; break; }   // src/async_mcts.rs:241-244 (B10)
        if (ec != E_NONE) { val = ecode_value(ec); kind = LEAF_VALUE; ++n_term; break; }  // :246-249
        // best_child, src/node.rs:343-370
        const uint32_t nchild = (pr.z >> META_NCHILD_SHIFT) & 7u, cb = pr.w;
        const float sq = puct_sqrt_parent(ctr_n(pc));
        uint4 cr = make_uint4(NONE, 0u, 0u, 0u);
        float u = 0.0f;
        if ((uint32_t)sub < nchild) {
            // the child's record and (speculatively) its own counter are fetched together; only a link slot needs
            // the second, dependent fetch of the canonical node's counter (resolve(), src/node.rs:179-193)
            cr = t.rec[base + cb + sub];
            uint64_t cc = t.ctr[base + cb + sub];
            if (cr.x != NONE) cc = t.ctr[base + cr.x];
            u = puct(cc, __uint_as_float(cr.y), sq, sp.cpuct_f);
        }
        uint32_t best = 0;
        float bu = gshflf(u, 0);
#pragma unroll
        for (int j = 1; j < ACTIONS; ++j) {                     // max_by: later element wins unless earlier is Greater (C7)
            float uj = gshflf(u, j);
            if ((uint32_t)j < nchild && !(bu > uj)) { best = (uint32_t)j; bu = uj; }
        }
        ++n_depth;
        const uint32_t clink = gshfl(cr.x, (int)best), cmeta = gshfl(cr.z, (int)best), cprior = gshfl(cr.y, (int)best);
        const uint32_t cslot = cb + best;
        if (plen >= (uint32_t)PATH_CAP) {
            if (sub == 0) atomicOr(&t.err[ERR_PATH], 1u);
            kind = LEAF_NONE;
            break;
        }
        if (sub == 0) path[plen] = cur;                         // node_path.push, S3 / :270
        ++plen;
        if (clink != NONE) { cur = clink; ++depth; continue; }  // Exists(false): follow the link (S2: one level per iteration)
        if (cmeta & META_EXPANDED) { cur = cslot; ++depth; continue; }   // Exists(true)
        // PlaceHolder (:261-268, S3): expand it.  B1: play the child's own action.
        ulonglong2 ps = t.state[base + cur];
        uint64_t m2, t2;
        c4_play(ps.x, ps.y, (int)(cmeta & META_A_MASK), &m2, &t2);       // :284-287 (B5)
        uint32_t found, ins;
        hash_find(t, g, base, m2, t2, sub, &found, &ins);
        if (found != NONE) {                                    // upgrade -> Some(false): become a link (src/node.rs:285-289)
            if (sub == 0) t.rec[base + cslot].x = found;
            cur = found;
            ++n_link;
            continue;                                           // :297-298
        }
        uint32_t ec2;
        // the placeholder is visited right after the upgrade (:309): its counter becomes INIT + VISIT
        if (!node_upgrade(t, g, base, cslot, m2, t2, cprior, cmeta & META_A_MASK, ins, CTR_INIT + CTR_VISIT, len_g, sub,
                          &ec2)) {
            kind = LEAF_NONE;
            break;
        }
        ++n_exp;
        cur = cslot;
        if (ec2 != E_NONE) { val = ecode_value(ec2); kind = LEAF_VALUE; break; }   // S4 (A5)
        kind = LEAF_EVAL;                                       // :303-315: goes to the net
        leaf_m = m2;
        leaf_t = t2;
        break;
    }
    if (sub == 0) {
        t.leaf[g] = cur;
        t.leaf_kind[g] = kind;
        t.leaf_val[g] = val;
        t.path_len[g] = plen;
        uint64_t* st = t.stat + (size_t)g * ST_COUNT;
        st[ST_SIMS] += 1;
        st[ST_EXPANSIONS] += n_exp;
        st[ST_LINK_HITS] += n_link;
        st[ST_TERMINAL_HITS] += n_term;
        st[ST_DEPTH_SUM] += n_depth;
    }
    batch_append(t, eb, g, sub == 0 && kind == LEAF_EVAL, leaf_m, leaf_t);
}

// ---- mask/renormalise/store the prior (src/async_mcts.rs:317-353) + backup (:361-370) ----
// `seen`-style sharing of evaluations across trees: a tree whose row was really evaluated publishes (pi, v) under its
// state's key.  Bucket = one 64-byte line of 8 keys; lane j looks at way j, lane 0 claims the first empty way with a CAS
// (another inserter may have taken it: try the next empty one), then lanes 0..7 write the 32-byte payload.  Readers are
// k_dedup launches LATER on the same stream, so a claimed key always has its payload by the time it can be matched.
AZ_D void cache_insert(const EvalCache& ec, uint64_t m, uint64_t th, float pv, int sub) {
    if ((uint32_t)__popcll(m | th) > ec.max_stones) return;
    const unsigned long long key = c4_key(m, th) | ec.tag;
    const uint32_t bucket = (uint32_t)(mix64(key) >> 20) & ec.bmask;
    unsigned long long* keys = ec.key + (size_t)bucket * 8;
    uint32_t empties = gballot(keys[sub] == 0ull);
    uint32_t way = NONE;
    while (empties) {
        const uint32_t w = (uint32_t)__ffs((int)empties) - 1u;
        unsigned long long prev = 0ull;
        if (sub == 0) prev = atomicCAS(&keys[w], 0ull, key);
        prev = ((unsigned long long)gshfl((uint32_t)(prev >> 32), 0) << 32) | gshfl((uint32_t)prev, 0);
        if (prev == 0ull) { way = w; break; }
        if (prev == key) return;                     // already published
        empties &= empties - 1u;
    }
    if (way == NONE) return;                         // bucket full: not cached
    ec.pv[((size_t)bucket * 8 + way) * 8 + sub] = pv;
    if (sub == 0) atomicAdd(&ec.stat[DD_INSERTS], 1ull);
}

AZ_D void backup_body(const TreeDev& t, const EvalBatch& eb, const EvalCache& ec, int apply_only, int g, int sub) {
    const uint32_t kind = t.leaf_kind[g];
    if (kind == LEAF_NONE) return;
    const size_t base = (size_t)g * t.R;
    const uint32_t leaf = t.leaf[g];
    float val;
    if (kind == LEAF_EVAL) {
        const int slot = t.slot_of[g];
        const ulonglong2 s = t.state[base + leaf];
        float pv;                                   // lanes 0..6: pi[sub], lane 7: v
        if (eb.src) {
            const uint32_t src = eb.src[slot];
            if (src & SRC_CACHE) {
                pv = ec.pv[(size_t)(src & SRC_INDEX) * 8 + sub];
            } else {
                const uint32_t u = (src & SRC_TABLE) ? eb.tuniq[src & SRC_INDEX] : src;
                pv = eb.upi[(size_t)u * 8 + sub];
                if (!(src & SRC_TABLE) && ec.key) cache_insert(ec, s.x, s.y, pv, sub);
            }
        } else {
            pv = eb.pi[(size_t)slot * 8 + sub];
        }
        float p = sub < ACTIONS ? pv : 0.0f;
        const float v = gshflf(pv, 7);
        if (t.log_cap > 0) {
            uint32_t n = t.log_len[g];
            if (n < (uint32_t)t.log_cap) {
                size_t li = (size_t)g * t.log_cap + n;
                if (sub < ACTIONS) t.log_pi[li * 7 + sub] = p;
                if (sub == 0) { t.log_v[li] = v; t.log_state[li] = s; }
            }
            if (sub == 0) t.log_len[g] = n + 1;
        }
        const uint32_t vm = c4_valid_mask(s.x, s.y);
        const bool valid = sub < ACTIONS && ((vm >> sub) & 1u);
        if (!valid) p = 0.0f;                                       // :322-326
        float sum = 0.0f;
#pragma unroll
        for (int a = 0; a < ACTIONS; ++a) sum = __fadd_rn(sum, gshflf(p, a));   // :328 (sequential, C10)
        if (sum > 0.0f) {
            p = __fdiv_rn(p, sum);                                  // :331
        } else {
            p = __fadd_rn(p, valid ? 1.0f : 0.0f);                  // :340-342
            float s2 = 0.0f;
#pragma unroll
            for (int a = 0; a < ACTIONS; ++a) s2 = __fadd_rn(s2, gshflf(p, a));
            p = __fdiv_rn(p, s2);                                   // :344
        }
        const uint4 lr = t.rec[base + leaf];
        const uint32_t nchild = (lr.z >> META_NCHILD_SHIFT) & 7u, cb = lr.w;
        const uint32_t myact = (uint32_t)sub < nchild ? nth_set_bit(vm, (uint32_t)sub) : 0u;
        const float pa = gshflf(p, (int)myact);
        if ((uint32_t)sub < nchild) t.rec[base + cb + sub].y = __float_as_uint(pa);   // set_policy, :348
        if (sub == 0) {
            t.rec[base + leaf].z = lr.z | META_HAS_PRIOR;
            t.stat[(size_t)g * ST_COUNT + ST_LEAF_EVALS] += 1;
        }
        val = -v;                                                   // :353 (C9)
    } else {
        val = t.leaf_val[g];
    }
    if (apply_only) return;
    // unvisit() leaf -> root along node_path; B2: the sign alternates toward the root.
    // A Connect Four line never repeats a node, so the lanes update distinct counters.
    const uint32_t plen = t.path_len[g];
    const uint32_t* path = t.path + (size_t)g * PATH_CAP;
    for (uint32_t i = (uint32_t)sub; i <= plen; i += LANES) {
        uint32_t node = i == 0 ? leaf : path[plen - i];
        float x = (i & 1u) ? -val : val;
        t.ctr[base + node] -= ctr_unvisit_delta(x);                 // src/node.rs:83-92
    }
}

// ---- get_action_prob epilogue (src/async_mcts.rs:84-114): counts -> pi ----------------------
struct RootPolicy {
    float pi;        // this lane's action (sub < 7)
    uint32_t count;
    float q;
};
AZ_D RootPolicy root_policy(const TreeDev& t, int g, int sub, float temp, uint64_t seed, uint64_t game_id, uint64_t ply) {
    const size_t base = (size_t)g * t.R;
    const uint4 pr = t.rec[base + t.root[g]];
    const uint32_t nchild = (pr.z >> META_NCHILD_SHIFT) & 7u, cb = pr.w;
    uint32_t ca = 0, cn = 0;
    float cq = 0.0f;
    if ((uint32_t)sub < nchild) {
        uint4 cr = t.rec[base + cb + sub];
        uint32_t r = cr.x != NONE ? cr.x : cb + (uint32_t)sub;
        uint64_t cc = t.ctr[base + r];
        ca = cr.z & META_A_MASK;                                    // B3: the slot's own action
        cn = ctr_n(cc);
        cq = ctr_q(cc);
    }
    RootPolicy out{0.0f, 0u, 0.0f};
#pragma unroll
    for (int j = 0; j < ACTIONS; ++j) {                             // counts[a] = n, :88-94
        uint32_t aj = gshfl(ca, j), nj = gshfl(cn, j);
        float qj = gshflf(cq, j);
        if ((uint32_t)j < nchild && aj == (uint32_t)sub) { out.count = nj; out.q = qj; }
    }
    if (temp == 0.0f) {                                             // :97-107
        uint32_t mx = 0;
#pragma unroll
        for (int a = 0; a < ACTIONS; ++a) { uint32_t ca2 = gshfl(out.count, a); mx = ca2 > mx ? ca2 : mx; }
        uint32_t ties = gballot(sub < ACTIONS && out.count == mx) & 0x7Fu;
        uint64_t r = rng_draw(seed, game_id, ply, RNG_TIEBREAK);
        uint32_t pick = nth_set_bit(ties, rng_choose(r, (uint32_t)__popc(ties)));
        out.pi = ((uint32_t)sub == pick) ? 1.0f : 0.0f;
    } else {                                                        // S6 (A7): counts^(1/temp) / sum
        float inv_t = __fdiv_rn(1.0f, temp);
        float x = (inv_t == 1.0f) ? (float)out.count : powf((float)out.count, inv_t);   // :109
        if (sub >= ACTIONS) x = 0.0f;
        float sum = 0.0f;
#pragma unroll
        for (int a = 0; a < ACTIONS; ++a) sum = __fadd_rn(sum, gshflf(x, a));           // :110
        out.pi = __fdiv_rn(x, sum);
    }
    return out;
}

__global__ __launch_bounds__(64) void k_select(TreeDev t, EvalBatch eb, SearchParams sp) {
    const int tid = blockIdx.x * 64 + threadIdx.x;
    const int g = tid >> 3, sub = tid & 7;
    if (g >= t.G) return;
    select_body(t, eb, sp, g, sub);
}

__global__ __launch_bounds__(64) void k_backup(TreeDev t, EvalBatch eb, EvalCache ec, int apply_only) {
    const int tid = blockIdx.x * 64 + threadIdx.x;
    const int g = tid >> 3, sub = tid & 7;
    if (tid == 0) { *eb.n = 0; if (eb.un) *eb.un = 0; }   // the batch has been consumed (nothing in this kernel reads the counts)
    if (g >= t.G) return;
    backup_body(t, eb, ec, apply_only, g, sub);
}

// backup of simulation i and select of simulation i+1 in one launch: both belong to the same 8 lanes of the same tree and
// nothing else touches that tree in between.  The leaf of i+1 goes into the OTHER eval batch (eb_next; its count was
// zeroed by the previous launch, this one zeroes eb_prev's), so the two ping-pong.
__global__ __launch_bounds__(64) void k_backup_select(TreeDev t, EvalBatch eb_prev, EvalBatch eb_next, EvalCache ec,
                                                      SearchParams sp, int apply_only) {
    const int tid = blockIdx.x * 64 + threadIdx.x;
    const int g = tid >> 3, sub = tid & 7;
    if (tid == 0) { *eb_prev.n = 0; if (eb_prev.un) *eb_prev.un = 0; }
    if (g >= t.G) return;
    backup_body(t, eb_prev, ec, apply_only, g, sub);
    // the counters this tree's other lanes just wrote are read by the selection below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    select_body(t, eb_next, sp, g, sub);
}

// ---- leaf de-duplication: one thread per requested row ------------------------------------------------------------------
// 1. evaluation cache: the row's bucket is one 64-byte line of 8 keys; a match ends the row (src = cache entry).
// 2. election table (open addressing, linear probe): the first row to CAS its key in wins and takes the next row of the
//    unique batch (one atomicAdd per wave); later rows with the same key point at the winner's slot.  A slot whose epoch
//    is not this launch's counts as empty, so the table is never cleared between launches (the caller clears it when the
//    15-bit epoch wraps).
__global__ __launch_bounds__(256) void k_dedup(EvalBatch eb, EvalCache ec, uint32_t epoch) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = *eb.n;
    const int lane = (int)(threadIdx.x & 63);
    bool winner = false, hit = false, dup = false;
    ulonglong2 s = make_ulonglong2(0ull, 0ull);
    if (r < n) {
        s = eb.state[r];
        const unsigned long long key = c4_key(s.x, s.y);
        if (ec.key) {
            const unsigned long long ck = key | ec.tag;
            const uint32_t bucket = (uint32_t)(mix64(ck) >> 20) & ec.bmask;
            const ulonglong4* kp = (const ulonglong4*)(ec.key + (size_t)bucket * 8);
            const ulonglong4 k0 = kp[0], k1 = kp[1];
            const int way = k0.x == ck ? 0 : k0.y == ck ? 1 : k0.z == ck ? 2 : k0.w == ck ? 3 :
                            k1.x == ck ? 4 : k1.y == ck ? 5 : k1.z == ck ? 6 : k1.w == ck ? 7 : -1;
            if (way >= 0) { hit = true; eb.src[r] = SRC_CACHE | (bucket * 8u + (uint32_t)way); }
        }
        if (!hit) {
            const unsigned long long mine = key | ((unsigned long long)epoch << 49);
            uint32_t pos = (uint32_t)(mix64(key) >> 24) & eb.tmask;
            for (;;) {
                unsigned long long cur = eb.tkey[pos];
                if ((cur >> 49) != (unsigned long long)epoch) {          // empty or stale: try to take it
                    const unsigned long long prev = atomicCAS(&eb.tkey[pos], cur, mine);
                    if (prev == cur) { winner = true; break; }
                    cur = prev;                                          // somebody else took it first
                    if ((cur >> 49) != (unsigned long long)epoch) continue;   // (a stale value replaced by another stale one cannot happen; retry anyway)
                }
                if (cur == mine) { dup = true; eb.src[r] = SRC_TABLE | pos; break; }
                pos = (pos + 1u) & eb.tmask;
            }
            if (winner) eb.src[r] = pos;            // provisional: replaced by the unique row below
        }
    }
    // unique rows: one atomicAdd per wave
    const unsigned long long wm = __ballot(winner);
    if (wm) {
        const int leader = __ffsll((long long)wm) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(eb.un, (uint32_t)__popcll(wm));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (winner) {
            const uint32_t u = base + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull));
            eb.tuniq[eb.src[r]] = u;
            eb.src[r] = u;
            eb.ustate[u] = s;
        }
    }
    if (ec.stat) {
        const unsigned long long hm = __ballot(hit), dm = __ballot(dup), am = __ballot(r < n);
        if (am && lane == __ffsll((long long)am) - 1) {
            atomicAdd(&ec.stat[DD_REQUESTED], (unsigned long long)__popcll(am));
            if (wm) atomicAdd(&ec.stat[DD_EXECUTED], (unsigned long long)__popcll(wm));
            if (hm) atomicAdd(&ec.stat[DD_CACHE_HITS], (unsigned long long)__popcll(hm));
            if (dm) atomicAdd(&ec.stat[DD_BATCH_DUPS], (unsigned long long)__popcll(dm));
        }
    }
}

__global__ __launch_bounds__(64) void k_root_policy(TreeDev t, float temp, uint64_t seed, uint64_t first_game_id,
                                                    float* pi, uint16_t* counts, float* q) {
    int tid = blockIdx.x * 64 + threadIdx.x;
    int g = tid >> 3, sub = tid & 7;
    if (g >= t.G || !t.active[g]) return;
    const ulonglong2 s = t.state[(size_t)g * t.R + t.root[g]];
    RootPolicy rp = root_policy(t, g, sub, temp, seed, first_game_id + (uint64_t)g, (uint64_t)__popcll(s.x | s.y));
    if (sub < ACTIONS) {
        pi[(size_t)g * 7 + sub] = rp.pi;
        if (counts) counts[(size_t)g * 7 + sub] = (uint16_t)rp.count;
        if (q) q[(size_t)g * 7 + sub] = rp.q;
    }
}

// ---- Coach::execute_episode, one ply for every slot (src/coach.rs:118-156) -------------------
__global__ __launch_bounds__(64) void k_selfplay_move(TreeDev t, GamesDev gd, SelfplayMoveParams mp) {
    int tid = blockIdx.x * 64 + threadIdx.x;
    int g = tid >> 3, sub = tid & 7;
    if (g >= t.G) return;
    const int gi = gd.gid[g];
    if (gi < 0 || !t.active[g]) return;
    const int ply = gd.ply[g];
    const int8_t player = gd.player[g];
    const ulonglong2 s = gd.state[g];
    const uint64_t game_id = mp.first_game_id + (uint64_t)gi;
    const float temp = (ply + 1 < mp.temp_threshold) ? 1.0f : 0.0f;         // :122-126 (episode_step = ply + 1)
    RootPolicy rp = root_policy(t, g, sub, temp, mp.seed, game_id, (uint64_t)ply);   // :128
    const size_t so = (size_t)gi * 42 + ply;
    if (sub < ACTIONS) gd.smp_pi[so * 7 + sub] = rp.pi;                     // :130-135 (symmetries regenerated at emit)
    // choose_weighted, :137-138
    float w[ACTIONS];
#pragma unroll
    for (int a = 0; a < ACTIONS; ++a) w[a] = gshflf(rp.pi, a);
    float total = 0.0f;
#pragma unroll
    for (int a = 0; a < ACTIONS; ++a) total = __fadd_rn(total, w[a]);
    const uint64_t r = rng_draw(mp.seed, game_id, (uint64_t)ply, RNG_MOVE);
    const float uu = (float)(uint32_t)(r >> 40) * (1.0f / 16777216.0f);
    const float target = __fmul_rn(uu, total);
    float acc = 0.0f;
    int action = -1, last = -1;
#pragma unroll
    for (int a = 0; a < ACTIONS; ++a) {
        if (w[a] > 0.0f) {
            acc = __fadd_rn(acc, w[a]);
            last = a;
            if (action < 0 && target < acc) action = a;
        }
    }
    if (action < 0) action = last;
    if (action < 0) action = 0;
    uint64_t m2, t2;
    c4_play(s.x, s.y, action, &m2, &t2);                                    // :140-142
    const uint32_t ec = c4_ecode(m2, t2);                                   // r = get_game_ended(cur_player), :144
    if (sub == 0) {
        gd.smp_state[so] = s;
        gd.smp_player[so] = player;
        gd.moves[so] = (uint8_t)action;
        if (ec != E_NONE) {
            // canonical ended(s') = -e: -1 when the side to move has lost, DRAW_EPS on a full board
            gd.g_result[gi] = -ecode_value(ec);
            gd.g_final_player[gi] = (int8_t)-player;
            gd.g_len[gi] = ply + 1;
            atomicAdd(&gd.counters[1], 1u);
            int next = -1;
            if (mp.refill) {
                uint32_t nx = atomicAdd(&gd.counters[0], 1u);
                if (nx < (uint32_t)gd.n_games) next = (int)nx;
            }
            gd.gid[g] = next;
            if (next >= 0) {
                gd.state[g] = make_ulonglong2(0ull, 0ull);
                gd.player[g] = 1;
                gd.ply[g] = 0;
                gd.need_reset[g] = 1;
            } else {
                t.active[g] = 0;
                atomicSub(&gd.counters[2], 1u);
            }
        } else {
            gd.state[g] = make_ulonglong2(m2, t2);
            gd.player[g] = (int8_t)-player;
            gd.ply[g] = ply + 1;
        }
    }
}

// ---- training tuples (TrainingSample, src/nnet.rs:22-27; z per B4, src/coach.rs:146-154) -----
__global__ __launch_bounds__(256) void k_emit_samples(GamesDev gd, const int64_t* offsets, int symmetries,
                                                      ulonglong2* out_states, float* out_boards, float* out_pis,
                                                      float* out_zs) {
    const int gi = blockIdx.x;
    const int len = gd.g_len[gi];
    const int nsym = symmetries ? 2 : 1;
    const float r = gd.g_result[gi];
    const int8_t fin = gd.g_final_player[gi];
    for (int item = threadIdx.x; item < len * nsym * 84; item += blockDim.x) {
        const int f = item % 84, rest = item / 84;
        const int sym = rest % nsym, ply = rest / nsym;
        const size_t so = (size_t)gi * 42 + ply;
        const int64_t o = (offsets[gi] + ply) * nsym + sym;
        ulonglong2 s = gd.smp_state[so];
        if (sym) s = make_ulonglong2(c4_mirror(s.x), c4_mirror(s.y));          // get_symmetries, connect_four_game.rs:205-211
        if (out_boards) out_boards[o * 84 + f] = c4_feature(s.x, s.y, f / 42, (f % 42) / 7, f % 7);
        if (f < 7) out_pis[o * 7 + f] = gd.smp_pi[so * 7 + (sym ? 6 - f : f)];
        if (f == 7) out_zs[o] = __fmul_rn(r, gd.smp_player[so] == fin ? 1.0f : -1.0f);   // B4
        if (f == 8 && out_states) out_states[o] = s;
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
    tn.active[g] = alive && mover == 0;
    to.active[g] = alive && mover == 1;
}

// The searching tree's owner plays argmax(get_action_prob(s, temp = 0)) (src/coach.rs:356-372).
__global__ __launch_bounds__(64) void k_arena_move(TreeDev t, ArenaDev ad, uint64_t seed) {
    int tid = blockIdx.x * 64 + threadIdx.x;
    int g = tid >> 3, sub = tid & 7;
    if (g >= t.G || !t.active[g]) return;
    const ulonglong2 s = ad.state[g];
    const int8_t player = ad.player[g];
    RootPolicy rp = root_policy(t, g, sub, 0.0f, seed, (uint64_t)(ad.first + g), (uint64_t)__popcll(s.x | s.y));
    // argmax with max_by (last max) over the one-hot pi = the index of the 1
    const uint32_t hot = gballot(sub < ACTIONS && rp.pi == 1.0f) & 0x7Fu;
    const int action = hot ? (31 - __clz((int)hot)) : 0;
    const bool valid = (c4_valid_mask(s.x, s.y) >> action) & 1u;      // src/arena.rs:29-35
    uint64_t m2, t2;
    c4_play(s.x, s.y, action, &m2, &t2);
    const uint32_t ec = c4_ecode(m2, t2);
    if (sub == 0) {
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
            ad.state[g] = make_ulonglong2(m2, t2);
            ad.player[g] = (int8_t)-player;
        }
    }
}

__global__ void k_sync_active(TreeDev t, GamesDev gd) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < t.G) t.active[g] = gd.gid[g] >= 0 ? 1 : 0;
}

// ---- launchers --------------------------------------------------------------------------------
static inline int group_blocks(int G) { return (G * LANES + 63) / 64; }

void launch_reset_trees(const TreeDev& t, const uint8_t* flags, hipStream_t s, const ulonglong2* roots) {
    hipLaunchKernelGGL(k_reset_trees, dim3(t.G), dim3(256), 0, s, t, flags, const_cast<uint8_t*>(flags), roots);
}
void launch_root_prepare(const TreeDev& t, const EvalBatch& eb, const ulonglong2* root_states, hipStream_t s) {
    hipLaunchKernelGGL(k_root_prepare, dim3(group_blocks(t.G)), dim3(64), 0, s, t, eb, root_states);
}
void launch_select(const TreeDev& t, const EvalBatch& eb, SearchParams sp, hipStream_t s) {
    hipLaunchKernelGGL(k_select, dim3(group_blocks(t.G)), dim3(64), 0, s, t, eb, sp);
}
void launch_backup(const TreeDev& t, const EvalBatch& eb, const EvalCache& ec, int apply_only, hipStream_t s) {
    hipLaunchKernelGGL(k_backup, dim3(group_blocks(t.G)), dim3(64), 0, s, t, eb, ec, apply_only);
}
void launch_backup_select(const TreeDev& t, const EvalBatch& eb_prev, const EvalBatch& eb_next, const EvalCache& ec, SearchParams sp,
                          int apply_only, hipStream_t s) {
    hipLaunchKernelGGL(k_backup_select, dim3(group_blocks(t.G)), dim3(64), 0, s, t, eb_prev, eb_next, ec, sp, apply_only);
}
void launch_dedup(const EvalBatch& eb, const EvalCache& ec, uint32_t epoch, hipStream_t s) {
    hipLaunchKernelGGL(k_dedup, dim3((eb.cap + 255) / 256), dim3(256), 0, s, eb, ec, epoch);
}
void launch_root_policy(const TreeDev& t, float temp, uint64_t seed, uint64_t first_game_id, float* pi,
                        uint16_t* counts, float* q, hipStream_t s) {
    hipLaunchKernelGGL(k_root_policy, dim3(group_blocks(t.G)), dim3(64), 0, s, t, temp, seed, first_game_id, pi, counts, q);
}
void launch_selfplay_move(const TreeDev& t, const GamesDev& gd, SelfplayMoveParams mp, hipStream_t s) {
    hipLaunchKernelGGL(k_selfplay_move, dim3(group_blocks(t.G)), dim3(64), 0, s, t, gd, mp);
}
void launch_selfplay_sync_active(const TreeDev& t, const GamesDev& gd, hipStream_t s) {
    hipLaunchKernelGGL(k_sync_active, dim3((t.G + 255) / 256), dim3(256), 0, s, t, gd);
}
void launch_arena_sync(const TreeDev& t_new, const TreeDev& t_old, const ArenaDev& ad, hipStream_t s) {
    hipLaunchKernelGGL(k_arena_sync, dim3((ad.G + 255) / 256), dim3(256), 0, s, t_new, t_old, ad);
}
void launch_arena_move(const TreeDev& t, const ArenaDev& ad, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(k_arena_move, dim3(group_blocks(t.G)), dim3(64), 0, s, t, ad, seed);
}
void launch_emit_samples(const GamesDev& gd, const int64_t* offsets, int symmetries, ulonglong2* out_states,
                         float* out_boards, float* out_pis, float* out_zs, hipStream_t s) {
    hipLaunchKernelGGL(k_emit_samples, dim3(gd.n_games), dim3(256), 0, s, gd, offsets, symmetries, out_states,
                       out_boards, out_pis, out_zs);
}

}  // namespace az
