// az_common.h -- device/host helpers shared by the engine's HIP kernels.
//
// Everything here is integer work or IEEE f32 in a fixed operation order: the
// engine is compiled with -ffp-contract=off and uses the *_rn intrinsics for the
// PUCT term so results match the reference's arithmetic (src/node.rs:51-92,
// :343-357) bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AZ_HD __host__ __device__ __forceinline__
#define AZ_D __device__ __forceinline__

namespace az {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int PATH_CAP = 48;          // node_path entries per simulation (a Connect Four line has at most 42 plies)

// ---- node meta word (the record itself: az_tree.h) ----------------------------
// bits 0-2 a (src/node.rs:19) | 3-5 nchild | 6 expanded (mu.s is Some) | 7-8 ecode (e, src/node.rs:20) | 9 has_prior (mu.p is Some)
// | 10 locked (the slot's AtomicBool, src/node.rs:130, :328-341: set from the upgrade of a placeholder until its prior is stored;
//   only ever observed with several simulations in flight per tree)
constexpr uint32_t META_A_MASK = 7u;
constexpr uint32_t META_NCHILD_SHIFT = 3;
constexpr uint32_t META_EXPANDED = 1u << 6;
constexpr uint32_t META_ECODE_SHIFT = 7;
constexpr uint32_t META_HAS_PRIOR = 1u << 9;
constexpr uint32_t META_LOCKED = 1u << 10;

// ecode <-> e (C9: e = -get_game_ended(1) of the canonical state)
constexpr uint32_t E_NONE = 0, E_PLUS1 = 1, E_MINUS1 = 2, E_DRAW = 3;
AZ_HD float ecode_value(uint32_t c) {
    return c == E_PLUS1 ? 1.0f : (c == E_MINUS1 ? -1.0f : (c == E_DRAW ? -1e-4f : 0.0f));
}

// ---- packed win counter 0xWWWWWWWW_NNNN_VVVV, src/node.rs:17,36 ---------------
constexpr uint64_t CTR_INIT = 0x7FFFFFFF00000000ull;
constexpr uint64_t CTR_VISIT = 0x0000000000010001ull;   // src/node.rs:77-80
constexpr float WIN_SCALE = 100.0f;                     // src/node.rs:13
AZ_HD uint32_t ctr_n(uint64_t c) { return (uint32_t)((c >> 16) & 0xFFFFu); }
AZ_HD uint32_t ctr_vloss(uint64_t c) { return (uint32_t)(c & 0xFFFFu); }
// src/node.rs:83-92
AZ_D uint64_t ctr_unvisit_delta(float win_val) {
    uint32_t incr = (uint32_t)fabsf(__fmul_rn(WIN_SCALE, win_val));
    uint64_t d = (win_val < 0.0f) ? ((uint64_t)incr << 32) : ((uint64_t)(0xFFFFFFFFu - incr) << 32);
    return 1ull | d;
}
// src/node.rs:61-64 then :51-58
AZ_D float ctr_w(uint64_t c) {
    return __fdiv_rn((float)((int64_t)(c >> 32) - 0x7FFFFFFFll), WIN_SCALE);
}
AZ_D float ctr_q(uint64_t c) {
    uint32_t n = ctr_n(c);
    if (n == 0) return 0.0f;
    return __fdiv_rn(__fsub_rn(ctr_w(c), (float)ctr_vloss(c)), (float)n);
}
// src/node.rs:352-356 (C6): q + ((cpuct*p) * sqrt(N_parent + 1e-6)) / (1 + n_child)
AZ_D float puct(uint64_t child_ctr, float prior, float sqrt_parent, float cpuct_f) {
    float denom = (float)((ctr_n(child_ctr) + 1u) & 0xFFFFu);   // u16 arithmetic
    return __fadd_rn(ctr_q(child_ctr), __fdiv_rn(__fmul_rn(__fmul_rn(cpuct_f, prior), sqrt_parent), denom));
}
// f32::sqrt of the reference is IEEE (correctly rounded).  NOT __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS the HIP headers map it
// to __ocml_native_sqrt_f32 = a bare v_sqrt_f32 (1 ulp), which flips the arg-max between two children whose PUCT terms tie to the last
// bit (found by replaying the 4096 x 400 arena on the oracle: round 3).  __builtin_sqrtf is lowered to v_sqrt_f32 + the correction
// steps under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt, like the `/` behind __fdiv_rn.
AZ_D float puct_sqrt_parent(uint32_t parent_n) { return __builtin_sqrtf(__fadd_rn((float)parent_n, 1e-6f)); }

// ---- RNG of the build ------------------------------------------
AZ_HD uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
constexpr uint64_t RNG_TIEBREAK = 1, RNG_MOVE = 2, RNG_WEIGHTS = 3;
AZ_HD uint64_t rng_draw(uint64_t seed, uint64_t game_id, uint64_t ply, uint64_t purpose) {
    return mix64(mix64(mix64(mix64(seed) ^ game_id) ^ ply) ^ purpose);
}
AZ_HD uint32_t rng_choose(uint64_t r, uint32_t k) { return (uint32_t)(((r >> 32) * (uint64_t)k) >> 32); }
// trainer streams: batch row j of step t = rng_choose(rng_draw(seed, t, j, RNG_BATCH), n_samples);
// dropout: element idx of layer `layer` is KEPT iff the top 24 bits of its draw are below keep_prob * 2^24
constexpr uint64_t RNG_BATCH = 4;
AZ_HD bool dropout_keep(uint64_t mask_seed, uint32_t layer, uint64_t idx, uint32_t keep_thresh24) {
    const uint64_t r = mix64(mix64(mask_seed ^ ((uint64_t)(layer + 1) * 0xD1B54A32D192ED03ull)) ^ idx);
    return (uint32_t)(r >> 40) < keep_thresh24;
}

}  // namespace az
