// az_net.hip -- NNet::predict on the device (src/nnet.rs:40-44).
//
// The policy+value net of examples/connect_four_lib/connect_four_net.py:20-95 (repaired: 7 actions,
// [B,2,6,7] input) at inference: conv3x3(2->C,same) -> conv3x3(C->C,same) -> conv3x3(C->C,valid) ->
// conv3x3(C->C,valid) -> FC 6C->1024 -> FC 1024->512 -> {pi: FC 512->7 + softmax, v: FC 512->1 + tanh},
// BatchNorm folded into the weights, ReLU fused into each producer's epilogue, bf16 activations and
// weights with f32 accumulation.
//
// The only dense contraction of the hot path: conv2..4 and the two FCs run as ONE implicit-GEMM kernel
// on the CDNA4 matrix cores (v_mfma_f32_16x16x32_bf16), channels-last activations so that a K-step of
// 64 input channels of one filter tap is one contiguous 128-byte row segment; conv1 (K = 18) reads the
// bitboards directly (to_features fused, connect_four_game.rs:219-237) on the VALU.
#include "az_net.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace az {

constexpr int ACTIONS = ConnectFour::ACTIONS;      // the net is Connect Four's NNet (connect_four_net.py): 7 policy outputs

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- fixtures ------------------------------------------------------------------------------------
// DumbConnectFourNnet (examples/connect_four.rs:34-41, S9): pi = 1/width, v = +1; or the hash fixture.
__global__ void k_net_fixture(EvalBatch eb, int kind, uint64_t salt) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *eb.n) return;
    float pi[ACTIONS], v;
    if (kind == 0) {
#pragma unroll
        for (int a = 0; a < ACTIONS; ++a) pi[a] = __fdiv_rn(1.0f, 7.0f);
        v = 1.0f;
    } else {
        ulonglong2 s = eb.state[i];
        hashnet_eval(s.x, s.y, salt, pi, &v);
    }
#pragma unroll
    for (int a = 0; a < ACTIONS; ++a) eb.pi[(size_t)i * 8 + a] = pi[a];
    eb.pi[(size_t)i * 8 + 7] = v;            // (pi, v) as one 32-byte row: what the backup lanes read
    eb.v[i] = v;
}

void launch_net_fixture(const EvalBatch& eb, int kind, uint64_t salt, hipStream_t s) {
    hipLaunchKernelGGL(k_net_fixture, dim3((eb.cap + 255) / 256), dim3(256), 0, s, eb, kind, salt);
}

// ---- bf16 helpers ----------------------------------------------------------------------------------
static inline uint16_t f32_to_bf16_host(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x7FFFFFu)) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);                                       // round to nearest even
}
AZ_D uint32_t pack_bf16x2(float lo, float hi) {
    __bf16 a = (__bf16)lo, b = (__bf16)hi;       // v_cvt_pk_bf16_f32 (round to nearest even)
    return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
}

AZ_D uint32_t pack_f16x2(float lo, float hi) {
    _Float16 a = (_Float16)lo, b = (_Float16)hi;       // round to nearest even
    return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
}

// ---- conv1: 3x3 'same', 2 -> C, reading the bitboards (K4 + first conv fused) -------------------------
// out: act1 [n][8][9][C] bf16, interior rows 1..6 / cols 1..7 (the zero halo is conv2's 'same' padding).
// One WAVE = one board, a lane = 8 output channels (one 16-byte store per position; lanes c8 = lane, lane+64, ...).
// The stone tests are wave-uniform; the vector unit adds the <= 18 weight rows that hit a stone, in the fixed
// (ky, kx, plane) order, and packs.  (With one wave per POSITION the kernel was bound by the latency of the
// dependent bitboard load, 84 serial loads per wave: 150 us per 8192 boards.)
__global__ __launch_bounds__(256) void k_conv1(const EvalBatch eb, const float* __restrict__ w /*[18][C]*/,
                                               const float* __restrict__ bias /*[C]*/, uint16_t* __restrict__ out, int C) {
    // The folded weights (18 x C f32 = 36 KiB at C = 512) are staged in LDS once per block; blocks are persistent
    // (grid-stride over (leaf, position) items), so the tap loop reads LDS instead of L2.
    extern __shared__ __attribute__((aligned(16))) float w_lds[];        // [18][C] + [C] bias
    for (int i = threadIdx.x; i < 18 * C / 4; i += blockDim.x) ((float4*)w_lds)[i] = ((const float4*)w)[i];
    for (int i = threadIdx.x; i < C / 4; i += blockDim.x) ((float4*)(w_lds + 18 * C))[i] = ((const float4*)bias)[i];
    __syncthreads();
    const int cg = C / 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t boards = *eb.n;
    const uint32_t stride = gridDim.x * 4u;
    // one wave = one board (its 42 positions in turn): the dependent bitboard load is paid once per 42 KiB of output
    for (uint32_t b = blockIdx.x * 4u + (uint32_t)wave; b < boards; b += stride) {
        const ulonglong2 sv = eb.state[b];
        const uint32_t m_lo = __builtin_amdgcn_readfirstlane((uint32_t)sv.x), m_hi = __builtin_amdgcn_readfirstlane((uint32_t)(sv.x >> 32));
        const uint32_t t_lo = __builtin_amdgcn_readfirstlane((uint32_t)sv.y), t_hi = __builtin_amdgcn_readfirstlane((uint32_t)(sv.y >> 32));
        const uint64_t mine = ((uint64_t)m_hi << 32) | m_lo, theirs = ((uint64_t)t_hi << 32) | t_lo;
        for (int y = 0; y < 6; ++y)
            for (int x = 0; x < 7; ++x) {
                uint16_t* orow = out + (((size_t)b * 8 + (y + 1)) * 9 + (x + 1)) * (size_t)C;
                for (int c8 = lane; c8 < cg; c8 += 64) {
                    const float4* bp = (const float4*)(w_lds + 18 * C + c8 * 8);
                    float4 a0 = bp[0], a1 = bp[1];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int iy = y + ky - 1, ix = x + kx - 1;
                            if (iy < 0 || iy >= 6 || ix < 0 || ix >= 7) continue;
                            const uint64_t bit = 1ull << (ix * 7 + (5 - iy));
#pragma unroll
                            for (int ci = 0; ci < 2; ++ci) {
                                if (!((ci == 0 ? mine : theirs) & bit)) continue;
                                const float4* wp = (const float4*)(w_lds + ((ky * 3 + kx) * 2 + ci) * C + c8 * 8);
                                const float4 w0 = wp[0], w1 = wp[1];
                                a0.x += w0.x; a0.y += w0.y; a0.z += w0.z; a0.w += w0.w;
                                a1.x += w1.x; a1.y += w1.y; a1.z += w1.z; a1.w += w1.w;
                            }
                        }
                    uint4 o;
                    o.x = pack_bf16x2(fmaxf(a0.x, 0.0f), fmaxf(a0.y, 0.0f));
                    o.y = pack_bf16x2(fmaxf(a0.z, 0.0f), fmaxf(a0.w, 0.0f));
                    o.z = pack_bf16x2(fmaxf(a1.x, 0.0f), fmaxf(a1.y, 0.0f));
                    o.w = pack_bf16x2(fmaxf(a1.z, 0.0f), fmaxf(a1.w, 0.0f));
                    *(uint4*)(orow + c8 * 8) = o;
                }
            }
    }
}

// ---- conv1 as a TABLE ---------------------------------------------------------------------------------------------
// conv1's output at a board position depends on nothing but its 3x3 neighbourhood: 9 cells x {empty or outside the board,
// mine, theirs} = 3^9 = 19683 patterns.  The table holds relu(conv1) for every pattern ([19683][C] bf16, 20 MB at C = 512,
// built once per weight upload with k_conv1's own summation order, so the rows are bit-identical to what k_conv1 writes),
// and conv2's image DMA gathers its 128-byte row segments straight from it (global_load_lds takes any per-lane source
// address): no conv1 launch, no act1 round trip through HBM, and the hot patterns (empty neighbourhoods) live in L2.
// pattern index = sum over taps t = ky*3 + kx of cell(y+ky-1, x+kx-1) * 3^t, cell = 0 empty/outside, 1 mine, 2 theirs.
constexpr int CONV1_PATTERNS = 19683;
AZ_HD uint32_t conv1_pattern(uint64_t mine, uint64_t theirs, int y /*row from the top*/, int x) {
    uint32_t idx = 0, mul = 1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = y + ky - 1, ix = x + kx - 1;
            if (iy >= 0 && iy < 6 && ix >= 0 && ix < 7) {
                const uint64_t bit = 1ull << (ix * 7 + (5 - iy));
                idx += ((mine & bit) ? 1u : (theirs & bit) ? 2u : 0u) * mul;
            }
            mul *= 3u;
        }
    return idx;
}
__global__ __launch_bounds__(64) void k_conv1_table(const float* __restrict__ w /*[18][C]*/, const float* __restrict__ bias /*[C]*/,
                                                    uint16_t* __restrict__ table /*[19683][C]*/, int C) {
    const int idx = blockIdx.x;
    int cell[9], r = idx;
#pragma unroll
    for (int t = 0; t < 9; ++t) { cell[t] = r % 3; r /= 3; }
    for (int c8 = threadIdx.x; c8 < C / 8; c8 += 64) {
        const float4* bp = (const float4*)(bias + c8 * 8);
        float4 a0 = bp[0], a1 = bp[1];
#pragma unroll
        for (int t = 0; t < 9; ++t)                    // (ky, kx) order, then plane 0 (mine) before plane 1: k_conv1's order
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                if (cell[t] != ci + 1) continue;
                const float4* wp = (const float4*)(w + (t * 2 + ci) * C + c8 * 8);
                const float4 w0 = wp[0], w1 = wp[1];
                a0.x += w0.x; a0.y += w0.y; a0.z += w0.z; a0.w += w0.w;
                a1.x += w1.x; a1.y += w1.y; a1.z += w1.z; a1.w += w1.w;
            }
        uint4 o;
        o.x = pack_bf16x2(fmaxf(a0.x, 0.0f), fmaxf(a0.y, 0.0f));
        o.y = pack_bf16x2(fmaxf(a0.z, 0.0f), fmaxf(a0.w, 0.0f));
        o.z = pack_bf16x2(fmaxf(a1.x, 0.0f), fmaxf(a1.y, 0.0f));
        o.w = pack_bf16x2(fmaxf(a1.z, 0.0f), fmaxf(a1.w, 0.0f));
        *(uint4*)(table + (size_t)idx * C + c8 * 8) = o;
    }
}

// ---- conv2 as a TABLE ---------------------------------------------------------------------------------------------
// conv2 is LINEAR in conv1's output, and conv1's output at a position is one of 19683 table rows (above).  So the
// contribution of filter tap t at a neighbour with pattern q is itself tabulated:  U[q][t][co] = sum_ci W2[co][t][ci] * T[q][ci]
// ([19683][9][C] f16, 181 MB at C = 512; built once per weight upload as ONE bf16 MFMA GEMM of 93 GFLOP, M = 19683 patterns,
// K = C, N = 9C, f32 accumulate), and conv2 at an output position is  relu(b + sum over its <= 9 in-board taps of U[pattern
// of that neighbour][t]): nine gathered 1-KiB rows and 9 x C f32 adds instead of a 4608-long dot product per channel --
// 198 of the net's 329 MFLOP per leaf are never executed.  Same network function; the rounding differs from the MFMA
// path's (f32 sums over ci rounded to f16 -- 2^-11, below the bf16 rounding of the activation that follows -- then summed over
// taps in f32), so it is a kernel set of its own ("conv2_table"), not bit-identical to the GEMM sets but equally
// batch-independent and held to the same tolerance against the torch reference.
// One wave per (board, board row): lanes 0..20 compute the patterns of the three rows of neighbours, then the wave walks the
// row's 7 output positions; a lane owns 8 channels (16-byte f16 loads, one 16-byte bf16 store).  The rows of pattern 0 (an
// empty neighbourhood -- the most frequent one by far) are staged in LDS once per block.

// The same gather with the CHANNELS split over the XCDs.  The 181 MB table does not fit the 8 x 4 MiB of L2, and in k_conv2_table
// every XCD gathers whole 1-KiB rows, so each L2 holds a random eighth of the hot rows and half of the gathered bytes come from
// beyond it (PMC: 177 of 354 KB per board).  Workgroup ids go round the XCDs, so here workgroup id handles the 64-channel slice
// id & 7 of its boards: an XCD only ever touches ITS 128-byte column of every row -- 22.6 MB, of which the rows that occur in play
// (a stone never floats over an empty cell: ~4000 of the 19683 patterns) are ~4.6 MB, about one L2.  One wave per (board, slice):
// lanes 0..41 compute the board's 42 patterns once, then six passes of one board row each, lane = (x, 16-byte chunk); the nine
// neighbour patterns come from the owning lanes by ds_bpermute.  Same per-channel summation order as k_conv2_table: bit-identical.
__global__ __launch_bounds__(256, 4) void k_conv2_table_x(const EvalBatch eb, const uint16_t* __restrict__ U /*[19683 + 1][9][C] f16, last row 0*/,
                                                       const float* __restrict__ bias /*[C]*/, uint16_t* __restrict__ out /*[n][42][C] bf16*/,
                                                       int C) {
    const int nsl = C / 64;                                   // channel slices (8 at C = 512: one per XCD)
    const int slice = blockIdx.x % nsl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = *eb.n;
    const int x = lane < 56 ? lane >> 3 : 6, chunk = lane & 7;      // lanes 56..63 shadow x = 6 and do not store
    const int coff = slice * 64 + chunk * 8;                  // first of this lane's 8 channels
    const float4 b0 = *(const float4*)(bias + coff), b1 = *(const float4*)(bias + coff + 4);
    const uint32_t stride = (gridDim.x / nsl) * 4u;
    const float one = 1.0f;                                   // v_fma_mix_f32's f32 multiplier, in an SGPR
    // bpermute source lane (x 4) of every tap's neighbour in board row y = 1 (rows shift by 7 lanes); taps outside the board read
    // lane 63, which holds the all-zero row appended to the table: every load is unconditional, and adding +0 changes no sum
    for (uint32_t b = (blockIdx.x / nsl) * 4u + wave; b < n; b += stride) {
        const ulonglong2 sv = eb.state[b];
        // byte offset of U[pattern of this lane's position][0][0]; lanes 42.. : the zero row
        const uint32_t mybase = (lane < 42 ? conv1_pattern(sv.x, sv.y, lane / 7, lane % 7) : (uint32_t)CONV1_PATTERNS) * (uint32_t)(18 * C);
#pragma unroll 1
        for (int y = 0; y < 6; ++y) {
            uint32_t base[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int qy = y + t / 3 - 1, qx = x + t % 3 - 1;
                const bool in = qy >= 0 && qy < 6 && qx >= 0 && qx < 7;
                base[t] = (uint32_t)__builtin_amdgcn_ds_bpermute((in ? qy * 7 + qx : 63) << 2, (int)mybase);
            }
            __builtin_amdgcn_sched_barrier(0);              // all nine permutes in flight before the first load waits for one
            uint4 u[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) u[t] = *(const uint4*)((const char*)U + (base[t] + (uint32_t)(t * 2 * C + coff * 2)));
            __builtin_amdgcn_sched_barrier(0);              // ... and all nine loads before the first add waits for one
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {                   // taps in (ky, kx) order: the fixed summation order of a row
                const uint32_t w[4] = {u[t].x, u[t].y, u[t].z, u[t].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // acc += (float)f16 as ONE v_fma_mix_f32 (exact convert, one rounding: the same bits as convert + add)
                    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * i]) : "v"(w[i]), "s"(one));
                    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * i + 1]) : "v"(w[i]), "s"(one));
                }
            }
            if (lane < 56) {
                uint4 o;
                o.x = pack_bf16x2(fmaxf(acc[0] + b0.x, 0.f), fmaxf(acc[1] + b0.y, 0.f));
                o.y = pack_bf16x2(fmaxf(acc[2] + b0.z, 0.f), fmaxf(acc[3] + b0.w, 0.f));
                o.z = pack_bf16x2(fmaxf(acc[4] + b1.x, 0.f), fmaxf(acc[5] + b1.y, 0.f));
                o.w = pack_bf16x2(fmaxf(acc[6] + b1.z, 0.f), fmaxf(acc[7] + b1.w, 0.f));
                *(uint4*)(out + ((size_t)b * 42 + (size_t)(y * 7 + x)) * C + coff) = o;
            }
        }
    }
}

// ---- implicit GEMM on MFMA: out[M,N] = relu(A_gather[M,K] * W[N,K]^T + bias) ---------------------------
// Row m = (sample b, output position (y,x)); K index = tap * cin + c with tap = ky*tap_w + kx reading the
// input at position (y+ky, x+kx) of an [in_h][in_w][in_c] channels-last image (the 'same' conv reads a
// zero-haloed image, the 'valid' convs an un-padded one; the FCs are a single tap over cin = K).
struct GemmDesc {
    const uint16_t* A;      // bf16 activations
    const uint16_t* W;      // bf16 [N][K]
    const float* bias;      // f32 [N]
    uint16_t* out;          // bf16 [M][N]
    const uint32_t* n_dev;  // samples in the batch (device)
    int rows_per_sample;    // out_h*out_w
    int out_w;
    int in_h, in_w, in_c;   // input image geometry (positions, channels per position)
    int tap_w;              // 3 for the convs, 1 for the FCs
    int cin;                // channels per tap
    int K, N;
    int relu;
    unsigned long long* dbg;   // diagnostic builds only (ABLATE == 3): per-block {shader cycles, 100 MHz ticks}
    const ulonglong2* states;  // k_conv_img2<.., true> only: the batch's canonical bitboards (A is then the conv1 table)
    int out_f16;               // k_gemm_mfma only: store the raw f32 accumulators as f16 (no bias, no ReLU): the conv2 table build
    // device-side hand-over between the small-batch kernel and the tiled kernels of a layer (both may be launched when the host's
    // estimate cannot tell; exactly one of them finds the batch's row count on its side of the line and runs):
    int m_min;                 // tiled kernels: do nothing when the batch has at most this many output rows (0 = always run)
    int m_max;                 // k_gemm_skinny: do nothing when the batch has more than this many output rows
    int m_hi;                  // k_gemm_ring_auto: do nothing when the batch has more than this many output rows (0 = no upper bound): the two ring
                               // families of a layer are both launched when the host's estimate cannot tell which one the batch needs
    unsigned long long* acct;  // k_conv3_auto: {rows, working launches} it has processed (device counters of the workspace)
    const uint16_t* c3tab;     // conv_valid_tile<.., PLANES>: the LDS image's cell maps (Conv3Tables, built by convnet_prepare's workspace)
    const uint16_t* Wp;        // k_conv3_pp: the layer's weights packed as LDS stage images (ConvNet::wp3), nullptr = not available
    const uint16_t* Wr;        // gemm_ring_body: the layer's weights as the ring's LDS stage images [N / 128][K / 64][128 rows][128 B] in K-step order
                               // (ConvNet::wr: one stage = 16 KiB of CONSECUTIVE global bytes instead of 128 rows K * 2 bytes apart), nullptr = read W
};

constexpr int GBM = 128, GBN = 128, GBK = 64;

template <int LAYER>   // distinct kernel symbol per layer so profiles name the dominant kernel (1 = conv2)
__global__ __launch_bounds__(256, 2) void k_gemm_mfma(const GemmDesc d) {
    // [buf][A tile 128 rows x 128 B | W tile 128 rows x 128 B]; 16-byte chunk c of row r lives at slot c ^ (r & 7)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][(GBM + GBN) * 128];
    const int M = (int)(*d.n_dev) * d.rows_per_sample;
    const int ntaps = d.K / d.cin;
    const int NT = d.N / GBN;
    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so the NT column tiles of one
    // row tile are placed on the same XCD back to back and re-read the activation tile from that XCD's L2.
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int ntile = j % NT, mtile = (j / NT) * 8 + xcd;
    const int m0 = mtile * GBM, n0 = ntile * GBN;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // staging map: 16-byte item i = q*256 + tid -> tile row i>>3, LDS slot i&7, logical chunk slot ^ (row&7)
    const int srow = tid >> 3;
    const int chunk = (tid & 7) ^ (srow & 7);
    auto row_off = [&](int q) -> uint32_t {
        int m = m0 + q * 32 + srow;
        m = m < M ? m : M - 1;
        const int b = m / d.rows_per_sample, r = m - b * d.rows_per_sample;
        const int y = r / d.out_w, x = r - y * d.out_w;
        return (uint32_t)(((b * d.in_h + y) * d.in_w + x) * d.in_c + chunk * 8);
    };
    const uint32_t a_off0 = row_off(0), a_off1 = row_off(1), a_off2 = row_off(2), a_off3 = row_off(3);
    const uint32_t b_off0 = (uint32_t)((n0 + srow) * d.K + chunk * 8), b_off1 = b_off0 + 32u * (uint32_t)d.K,
                   b_off2 = b_off0 + 64u * (uint32_t)d.K, b_off3 = b_off0 + 96u * (uint32_t)d.K;
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    // global -> registers for K-step kt (no lambdas / arrays: keeps everything in VGPRs, no scratch)
#define AZ_GLOAD(kt_)                                                                           \
    {                                                                                           \
        /* K order: channel block outer, filter tap inner -> the 9 taps of one 64-channel block re-read   \
           the same 128-B segments back to back (L2-resident) instead of sweeping the whole image per tap */ \
        const int cb_ = (kt_) / ntaps, tap = (kt_) - cb_ * ntaps;                               \
        const int c0 = cb_ * GBK, kk = tap * d.cin + c0;                                        \
        const int ky = tap / d.tap_w, kx = tap - ky * d.tap_w;                                  \
        const uint32_t toff = (uint32_t)((ky * d.in_w + kx) * d.in_c + c0);                     \
        ra0 = *(const uint4*)(d.A + a_off0 + toff); ra1 = *(const uint4*)(d.A + a_off1 + toff); \
        ra2 = *(const uint4*)(d.A + a_off2 + toff); ra3 = *(const uint4*)(d.A + a_off3 + toff); \
        rb0 = *(const uint4*)(d.W + b_off0 + kk); rb1 = *(const uint4*)(d.W + b_off1 + kk);     \
        rb2 = *(const uint4*)(d.W + b_off2 + kk); rb3 = *(const uint4*)(d.W + b_off3 + kk);     \
    }
#define AZ_SWRITE(buf_)                                                                         \
    {                                                                                           \
        unsigned char* sa_ = smem[buf_] + tid * 16;                                             \
        unsigned char* sb_ = sa_ + GBM * 128;                                                   \
        *(uint4*)(sa_) = ra0; *(uint4*)(sa_ + 4096) = ra1; *(uint4*)(sa_ + 8192) = ra2; *(uint4*)(sa_ + 12288) = ra3; \
        *(uint4*)(sb_) = rb0; *(uint4*)(sb_ + 4096) = rb1; *(uint4*)(sb_ + 8192) = rb2; *(uint4*)(sb_ + 12288) = rb3; \
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = d.K / GBK;
    const int frow = lane & 15, fq = lane >> 4, fsw = lane & 7;
    AZ_GLOAD(0);
    AZ_SWRITE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) AZ_GLOAD(kt + 1);
        const unsigned char* sA = smem[kt & 1];
        const unsigned char* sB = sA + GBM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + fq) ^ fsw) << 4;
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) fa[mt] = *(const bf16x8*)(sA + (wr * 64 + mt * 16 + frow) * 128 + coff);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) fb[nt] = *(const bf16x8*)(sB + (wc * 64 + nt * 16 + frow) * 128 + coff);
            // D[n][m] = sum_k W[n][k] * Act[m][k]: the weight fragment is the MFMA's A operand so that each lane
            // ends up with 4 consecutive output channels of one activation row (an 8-byte packed store).
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
        }
        if (kt + 1 < nk) AZ_SWRITE((kt + 1) & 1);
        __syncthreads();
    }
#undef AZ_GLOAD
#undef AZ_SWRITE
    // epilogue: + bias, ReLU, bf16, out[m][n .. n+3]  (out_f16: the raw sums as f16)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wc * 64 + nt * 16 + fq * 4;
        const float4 bv = d.out_f16 ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(d.bias + n);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + wr * 64 + mt * 16 + frow;
            if (m >= M) continue;
            float r0 = acc[mt][nt][0] + bv.x, r1 = acc[mt][nt][1] + bv.y, r2 = acc[mt][nt][2] + bv.z,
                  r3 = acc[mt][nt][3] + bv.w;
            if (d.relu) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); r2 = fmaxf(r2, 0.f); r3 = fmaxf(r3, 0.f); }
            uint2 o;
            if (d.out_f16) {
                o.x = pack_f16x2(r0, r1);
                o.y = pack_f16x2(r2, r3);
            } else {
                o.x = pack_bf16x2(r0, r1);
                o.y = pack_bf16x2(r2, r3);
            }
            *(uint2*)(d.out + (size_t)m * d.N + n) = o;
        }
    }
}

// ---- LDS-DMA issued from inline asm ---------------------------------------------------------------------------------------
// hipcc models `__builtin_amdgcn_global_load_lds` as a FLAT access that may touch LDS and global memory at once: while one is
// pending, EVERY s_waitcnt the compiler inserts is vmcnt(0) / lgkmcnt(0), so each MFMA cluster waits for fragment reads issued
// just before it for the NEXT cluster.  Issued from inline asm (saddr form: uniform base in SGPRs + a 32-bit lane offset, LDS
// base in M0) the compiler does not see the DMA: its own lgkmcnt waits are counted, and the kernels below wait for the DMA
// themselves (explicit vmcnt + barrier before the first read of a landed tile).
__device__ __forceinline__ void lds_dma16(const void* sbase /*uniform*/, uint32_t voff, uint32_t lds_addr /*uniform*/) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

// ---- 128x128 tile, LDS-DMA RING: the small-grid layers (fc1, fc2; conv4 on small batches) ---------------------------
// fc1 (M = leaves, N = 1024, K = 3072) has 176 tiles at 2700 leaves and fc2 88: fewer workgroups than CUs, so k_gemm_mfma
// runs one 4-wave workgroup per CU and every K-step exposes a full L2 / HBM round trip between its global loads and the
// barrier that publishes them (~1 us per step, 48 steps: the kernel is latency-bound at 19 % of the MFMA peak).  Same tile,
// same per-row K order (bit-identical), but the operands go global -> LDS by asm-issued LDS-DMA into a ring of NS stages
// (BM x 128 B of A + 16 KiB of W each; 128 KiB at NS = 4, BM = 128: the LDS an under-filled CU has to spare), NS-1 stages in
// flight, retired with a COUNTED s_waitcnt (never 0 inside the loop) and one raw barrier per step; both 32-deep halves' fragments
// are requested up front and the second half's reads sit between the first half's MFMAs.
// BM = 64 .. 192 rows per tile (k_gemm_ring_auto picks it on the device): a CU fetches its tiles L2 -> LDS at a bounded rate
// (~19-27 B/clk) whatever else it does, so idle CUs are idle fetch bandwidth and a second, nearly empty round of workgroups is a
// whole round of time -- the tile is the smallest whose grid still fits whole rounds of workgroup slots.
template <int NS, int BM>
__device__ __forceinline__ void gemm_ring_body(const GemmDesc& d, unsigned char* smem /*NS * (BM * 128 + 16384) bytes of LDS*/, const int M) {
    static_assert(BM % 32 == 0 && BM >= 64 && BM <= 192 && NS * (BM * 128 + 16384) <= 163840 && 2 * (BM / 32 + 4) <= 63, "tile");
    constexpr int MT = BM / 32;                       // 16-row tiles per wave (2 x 2 waves) = A pieces per wave
    constexpr int STAGE = BM * 128 + 16384;           // [A BM x 128 B | W 128 x 128 B]
    constexpr int NDMA = MT + 4;                      // DMA instructions per wave per stage
    const int ntaps = d.K / d.cin;
    const int NT = d.N / GBN;
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int ntile = j % NT, mtile = (j / NT) * 8 + xcd;
    const int m0 = mtile * BM, n0 = ntile * GBN;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // DMA map: piece q of wave w fills tile rows (q*4 + w)*8 .. +7 (A: q < MT, W: q < 4); lane -> row lane>>3, slot lane&7
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ lrow;
    uint32_t a_ob[MT];
#pragma unroll
    for (int q = 0; q < MT; ++q) {
        int m = m0 + (q * 4 + wave) * 8 + lrow;
        m = m < M ? m : M - 1;
        const int b = m / d.rows_per_sample, r = m - b * d.rows_per_sample;
        const int y = r / d.out_w, x = r - y * d.out_w;
        a_ob[q] = (uint32_t)(((b * d.in_h + y) * d.in_w + x) * d.in_c + chunk * 8) * 2u;
    }
    const bool packed = d.Wr != nullptr;
    const uint32_t b_ob = packed ? (uint32_t)(wave * 1024 + lane * 16)
                                 : (uint32_t)((n0 + wave * 8 + lrow) * d.K + chunk * 8) * 2u;      // W piece q: + q * w_stride on the SGPR base
    const size_t w_stride = packed ? (size_t)4096 : (size_t)64 * d.K;
    const char* wr_tile = (const char*)d.Wr + (size_t)ntile * (size_t)(d.K / GBK) * 16384;
    int ks_idx = 0;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)(smem + wave * 1024);
    // K-step walker (channel block outer, tap inner), scalars only
    int ks_tap = 0, ks_kx = 0;
    uint32_t ks_c0 = 0, ks_toff = 0, ks_kk = 0;
#define AZ_RDMA(buf_)                                                                                   \
    {                                                                                                   \
        const char* abase = (const char*)(d.A + ks_toff);                                               \
        const char* wbase = packed ? wr_tile + (size_t)ks_idx * 16384 : (const char*)(d.W + ks_kk);     \
        ++ks_idx;                                                                                       \
        const uint32_t la = lds0 + (buf_) * STAGE;                                                      \
        _Pragma("unroll") for (int q_ = 0; q_ < MT; ++q_) lds_dma16(abase, a_ob[q_], la + q_ * 4096);   \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) lds_dma16(wbase + q_ * w_stride, b_ob, la + BM * 128 + q_ * 4096); \
        ++ks_tap; ++ks_kx; ks_toff += (uint32_t)d.in_c; ks_kk += (uint32_t)d.cin;                       \
        if (ks_kx == d.tap_w) { ks_kx = 0; ks_toff += (uint32_t)((d.in_w - d.tap_w) * d.in_c); }        \
        if (ks_tap == ntaps) { ks_tap = 0; ks_kx = 0; ks_c0 += GBK; ks_toff = ks_c0; ks_kk = ks_c0; }   \
    }
    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = d.K / GBK;
    const int frow = lane & 15, fq = lane >> 4, fsw = lane & 7;
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nk) AZ_RDMA(st);
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt must have landed; up to NS-2 younger stages (NDMA instructions each) may stay in flight
        const int younger = nk - 1 - kt < NS - 2 ? nk - 1 - kt : NS - 2;
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // raw: every wave is also done with stage kt-1's buffer
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NS - 1 < nk) AZ_RDMA((kt + NS - 1) % NS);
        const unsigned char* sA = smem + (kt % NS) * STAGE;
        const unsigned char* sB = sA + BM * 128;
        // both 32-deep halves' fragments are requested up front: the second half's reads sit between the first half's MFMAs
        const int coff0 = ((0 + fq) ^ fsw) << 4, coff1 = ((4 + fq) ^ fsw) << 4;
        bf16x8 fa0[MT], fb0[4], fa1[MT], fb1[4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa0[mt] = *(const bf16x8*)(sA + (wr * (BM / 2) + mt * 16 + frow) * 128 + coff0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) fb0[nt] = *(const bf16x8*)(sB + (wc * 64 + nt * 16 + frow) * 128 + coff0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa1[mt] = *(const bf16x8*)(sA + (wr * (BM / 2) + mt * 16 + frow) * 128 + coff1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) fb1[nt] = *(const bf16x8*)(sB + (wc * 64 + nt * 16 + frow) * 128 + coff1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[nt], fa0[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[nt], fa1[mt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, MT + 4, 0);
#pragma unroll
        for (int g = 0; g < MT + 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8 * MT - (MT + 4), 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef AZ_RDMA
    // epilogue through LDS (the ring is dead): + bias, ReLU, bf16 as [BM rows][128 channels] with a 272-byte row stride, then whole
    // 256-byte row segments, 16 bytes per lane (instead of 8-byte stores in 32-byte pieces of 16 different rows)
    constexpr int EP_STRIDE = 272;
    static_assert(BM * EP_STRIDE <= NS * STAGE, "the output tile must fit the ring");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int nl = wc * 64 + nt * 16 + fq * 4;
        const float4 bv = *(const float4*)(d.bias + n0 + nl);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int ml = wr * (BM / 2) + mt * 16 + frow;
            float r0 = acc[mt][nt][0] + bv.x, r1 = acc[mt][nt][1] + bv.y, r2 = acc[mt][nt][2] + bv.z,
                  r3 = acc[mt][nt][3] + bv.w;
            if (d.relu) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); r2 = fmaxf(r2, 0.f); r3 = fmaxf(r3, 0.f); }
            uint2 o;
            o.x = pack_bf16x2(r0, r1);
            o.y = pack_bf16x2(r2, r3);
            *(uint2*)(smem + ml * EP_STRIDE + nl * 2) = o;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < BM * 16 / 256; ++it) {
        const int idx = it * 256 + tid;
        const int ml = idx >> 4, c = idx & 15;
        const int m = m0 + ml;
        if (m >= M) continue;
        *(uint4*)(d.out + (size_t)m * d.N + n0 + c * 8) = *(const uint4*)(smem + ml * EP_STRIDE + c * 16);
    }
}


// Tile rows chosen ON THE DEVICE from the batch's row count (the host only knows an estimate when it launches).  A workgroup's time
// grows with its tile, a launch's with its ROUNDS of workgroup slots (NS = 4: one workgroup per CU, 256 slots; NS = 2: two, 512), so
// the best tile is the smallest whose grid still fits whole rounds (measured: tools/ring_tiles.py, profiles/README.md: conv4 at 3072
// rows takes 120 us on 128-row tiles -- 576 workgroups, a second round for 64 of them -- and 78 us on 160-row tiles).
AZ_HD int ring_pick_bm(int M, int ncol, int ns, bool conv) {
    auto wgs = [&](int bm) { return (M + bm - 1) / bm * ncol; };
    if (ns == 4) return wgs(64) <= 256 ? 64 : wgs(96) <= 256 ? 96 : 128;
    if (!conv) return wgs(96) <= 512 ? 96 : 128;
    // conv4 (K = 9 * cin), two workgroups per CU: cost = one round of BM-row tiles x (whole rounds + the last, partly filled one:
    // workgroups alone on their CU finish in ~0.6 of a shared round)
    int best = 128;
    float best_cost = 1e30f;
    for (int bm = 96; bm <= 192; bm += 32) {
        const int w = wgs(bm), full = w / 512, rem = w % 512;
        const float cost = (10.f + 0.42f * (float)bm) * ((float)full + (rem == 0 ? 0.f : rem <= 256 ? 0.6f : 1.f));
        if (cost < best_cost) { best_cost = cost; best = bm; }
    }
    return best;
}
template <int LAYER, int NS>
__global__ __launch_bounds__(256, (NS <= 2 ? 2 : 1)) void k_gemm_ring_auto(const GemmDesc d) {
    constexpr int BMAX = NS == 2 ? 192 : 128;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * (BMAX * 128 + 16384)];
    const int M = (int)(*d.n_dev) * d.rows_per_sample;
    if (M <= d.m_min || (d.m_hi > 0 && M > d.m_hi)) return;      // another kernel launched beside this one takes the batch
    const int bm = __builtin_amdgcn_readfirstlane(ring_pick_bm(M, d.N / GBN, NS, d.tap_w > 1));
    if constexpr (NS == 2) {
        if (bm == 96) gemm_ring_body<2, 96>(d, smem, M);
        else if (bm == 160) gemm_ring_body<2, 160>(d, smem, M);
        else if (bm == 192) gemm_ring_body<2, 192>(d, smem, M);
        else gemm_ring_body<2, 128>(d, smem, M);
    } else {
        if (bm == 64) gemm_ring_body<4, 64>(d, smem, M);
        else if (bm == 96) gemm_ring_body<4, 96>(d, smem, M);
        else gemm_ring_body<4, 128>(d, smem, M);
    }
}

// ---- 256x256 tile variant for the big layers (conv2, conv3): LDS-DMA staging ------------------------------
// 8 waves (2 x 4), each wave a 128(m) x 64(n) sub-tile = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16.
// Both operands go global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write): one wave-instruction
// writes 1 KiB = 8 tile rows x 128 B linearly, so the XOR swizzle (chunk c of row r at slot c ^ (r&7)) is applied
// to the per-lane SOURCE address and again on the fragment reads.  Two 64 KiB LDS buffers; the next K-step's DMA is
// issued before this K-step's MFMAs and retired by vmcnt(0) + barrier at the end of the step.
[[maybe_unused]] constexpr int HBM_ = 256;
constexpr int HBN_ = 256;


// ---- conv2 as an IMAGE-RESIDENT implicit GEMM ----------------------------------------------------------------
// The 9 filter taps of one 64-channel block read overlapping shifted windows of the same activations.  With an M tile
// of 6 whole boards (252 output rows) the 64-channel slice of those boards is 6 x 42 x 128 B = 31.5 KiB: it is DMA'd
// into LDS ONCE per channel block (double-buffered, landing during the previous block's taps) and the A fragments of
// tap (ky,kx) are read from it at row m + (ky-1)*7 + (kx-1); out-of-board taps read a zero row ('same' padding).
// Only the weight tile (32 KiB) still streams every K-step, so L2->LDS traffic falls from 64 KiB to 35.5 KiB per
// K-step.  Same K order (channel block outer, tap inner) and per-row accumulation order as the other kernels.
constexpr int IMG_NB = 6, IMG_ROWS = IMG_NB * 42, IMG_ZERO_ROW = 252;


// ---- conv2 image-resident, TWO independent workgroups per CU ---------------------------------------------------------
// k_conv_img's 8 waves share one barrier, so the two waves of every SIMD run in lockstep: both in their MFMA clusters
// (contending for the matrix pipe), then both parked at the wait / barrier (pipe idle; SQ_WAIT_INST_ANY 42 %,
// SQ_WAIT_ANY 38 % of wave cycles).  Here the same per-wave work (128 rows x 64 columns, 8 x 4 accumulators) is packaged
// as workgroups of 4 waves -- tile = 6 boards x 128 channels, LDS = one image buffer (32 KiB) + two weight buffers
// (16 KiB each) = 64 KiB -- so two workgroups fit a CU and every SIMD holds one wave of each: their barriers are
// independent and their phases drift apart.  The price is a single image buffer (the image switch every 9 K-steps is
// exposed inside a workgroup and covered by the other one).  Same K order: bit-identical.
constexpr int HBN2_ = 128;


// ---- k_conv_img2 with the LDS-DMA issued from inline asm and a software-pipelined K-step (conv2 as the MFMA GEMM, "conv2_table" = 0) --
// Same tile, LDS layout, DMA maps and K order as k_conv_img2 (bit-identical).  What changes is what k_conv_valid_pipe changed for conv3:
// the DMA is invisible to the compiler (its lgkmcnt waits are counted), fragment reads for the next cluster sit between this
// cluster's MFMAs, and the step's first fragments are requested under the previous step's last cluster.  The two weight buffers
// make ONE barrier per step enough: W(k+1) streams into the other buffer for the whole of step k (issued at its top), the barrier
// behind cluster 3 says "W(k+1) has landed everywhere and nobody reads W(k) any more".
__device__ __forceinline__ void lds_dma16(const void* sbase, uint32_t voff, uint32_t lds_addr);
template <int LAYER, bool TABLE = false>
__global__ __launch_bounds__(256, 2) void k_conv_same_pipe(const GemmDesc d) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 32768];   // img | w[2] (16 KiB each)
    const int n_boards = (int)(*d.n_dev);
    const int M = n_boards * 42;
    const int C = d.cin;
    const int NT = d.N / HBN2_;
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int ntile = j % NT, mtile = (j / NT) * 8 + xcd;
    const int b0 = mtile * IMG_NB, n0 = ntile * HBN2_;
    if (b0 >= n_boards) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    if (tid < 8) *(uint4*)(smem + IMG_ZERO_ROW * 128 + tid * 16) = make_uint4(0, 0, 0, 0);
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ lrow;
    uint32_t i_ob[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int r = (q * 4 + wave) * 8 + lrow;
        r = r < IMG_ROWS ? r : IMG_ROWS - 1;
        int b = b0 + r / 42;
        b = b < n_boards ? b : n_boards - 1;
        const int p = r % 42, y = p / 7, x = p - y * 7;
        if constexpr (TABLE) {
            const ulonglong2 st = d.states[b];
            i_ob[q] = (conv1_pattern(st.x, st.y, y, x) * (uint32_t)C + (uint32_t)chunk * 8u) * 2u;
        } else {
            i_ob[q] = (uint32_t)(((b * 8 + y + 1) * 9 + x + 1) * C + chunk * 8) * 2u;
        }
    }
    const bool i_last_ok = (7 * 4 + wave) * 8 + lrow < IMG_ROWS;          // only piece 7 can run past row 251
    const uint32_t w_ob = (uint32_t)((n0 + wave * 8 + lrow) * d.K + chunk * 8) * 2u;
    const size_t w_stride = (size_t)64 * d.K;                              // 32 weight rows, in bytes
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const uint32_t lds_img = (uint32_t)(uintptr_t)(lds_ptr)(smem + wave * 1024);
    const uint32_t lds_w = (uint32_t)(uintptr_t)(lds_ptr)(smem + 32768 + wave * 1024);
#define AZ_CDMA_W(kk_, buf_)                                                                                 \
    {                                                                                                        \
        const char* wbase = (const char*)(d.W + (kk_));                                                      \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) lds_dma16(wbase + q_ * w_stride, w_ob, lds_w + (buf_) * 16384 + q_ * 4096); \
    }
#define AZ_CDMA_IMG(cb_)                                                                                     \
    {                                                                                                        \
        const char* ibase = (const char*)(d.A + (cb_) * 64);                                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < 7; ++q_) lds_dma16(ibase, i_ob[q_], lds_img + q_ * 4096);    \
        if (i_last_ok) lds_dma16(ibase, i_ob[7], lds_img + 7 * 4096);                                        \
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4, fsw = lane & 7;
    uint32_t rowmask[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int ml = wr * 128 + mt * 16 + frow;
        const int p = ml % 42, y = p / 7, x = p - y * 7;
        uint32_t mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
            if (ml < IMG_ROWS && iy >= 0 && iy < 6 && ix >= 0 && ix < 7) mask |= 1u << t;
        }
        rowmask[mt] = (uint32_t)ml | (mask << 16);
    }
    const int b_row0 = 32768 + (wc * 64 + frow) * 128;
    const int coffB0 = ((0 + fq) ^ fsw) << 4, coffB1 = ((4 + fq) ^ fsw) << 4;
    // tap_: 0..8, its image-row offset dtv_ = (ky-1)*7 + (kx-1)
#define AZ_CLDA(dst_, mt0_, ks_, tap_, dtv_)                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                       \
        const uint32_t rm_ = rowmask[(mt0_) + i_];                                                           \
        const int r_ = ((rm_ >> (16 + (tap_))) & 1u) ? (int)(rm_ & 0xFFFFu) + (dtv_) : IMG_ZERO_ROW;         \
        dst_[i_] = *(const bf16x8*)(smem + r_ * 128 + ((((ks_) * 4 + fq) ^ (r_ & 7)) << 4));                 \
    }
#define AZ_CLDB(dst_, buf_, coff_)                                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                         \
        dst_[i_] = *(const bf16x8*)(smem + (buf_) * 16384 + b_row0 + i_ * 2048 + (coff_));
#define AZ_CMMA(mt0_, fb_, fa_)                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                         \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                     \
            acc[(mt0_) + i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb_[j_], fa_[i_], acc[(mt0_) + i_][j_], 0, 0, 0);
#define AZ_CSB __builtin_amdgcn_sched_barrier(0)
#define AZ_CMIX(nmf_, rep_, tail_)                                                                           \
    {                                                                                                        \
        _Pragma("unroll") for (int g_ = 0; g_ < (rep_); ++g_) {                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, (nmf_), 0);                                          \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                               \
        }                                                                                                    \
        if ((tail_) > 0) __builtin_amdgcn_sched_group_barrier(0x008, (tail_), 0);                            \
    }
    const int ncb = C / 64;
    const int nk = ncb * 9;
    AZ_CDMA_W(0, 0);
    AZ_CDMA_IMG(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8 fbX[4], fbY[4], faX[4], faY[4];
    AZ_CLDB(fbX, 0, coffB0);
    AZ_CLDA(faX, 0, 0, 0, -8);
    int cb = 0, tap = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool sw = tap == 8;
        const int ntap = sw ? 0 : tap + 1, ncbi = sw ? cb + 1 : cb;
        const int ky = tap / 3, dt = (ky - 1) * 7 + (tap - ky * 3 - 1);
        const int nky = ntap / 3, ndt = (nky - 1) * 7 + (ntap - nky * 3 - 1);
        const int buf = kt & 1;
        const int kk = kt + 1 < nk ? ntap * C + ncbi * 64 : 8 * C + cb * 64;      // last step: re-fetch its own tile (unused)
        AZ_CDMA_W(kk, buf ^ 1);                                  // the other buffer: nobody has read it since the last barrier
        AZ_CLDB(fbY, buf, coffB1);
        AZ_CLDA(faY, 4, 0, tap, dt);
        AZ_CMMA(0, fbX, faX);
        AZ_CMIX(1, 8, 8);
        AZ_CSB;
        AZ_CLDA(faX, 0, 1, tap, dt);
        AZ_CMMA(4, fbX, faY);
        AZ_CMIX(2, 4, 8);
        AZ_CSB;
        AZ_CLDA(faY, 4, 1, tap, dt);
        AZ_CMMA(0, fbY, faX);
        AZ_CMIX(2, 4, 8);
        AZ_CSB;
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // my reads of W(k) and of this step's image rows are done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // W(k+1) has landed
        __builtin_amdgcn_s_barrier();
        AZ_CSB;
        if (sw && ncbi < ncb) {                                  // single image buffer: the switch is covered by the CU's other workgroup
            AZ_CDMA_IMG(ncbi);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        AZ_CSB;
        AZ_CLDB(fbX, buf ^ 1, coffB0);                           // next step's first fragments, under this step's last cluster
        AZ_CLDA(faX, 0, 0, ntap, ndt);
        AZ_CMMA(4, fbY, faY);
        AZ_CMIX(1, 8, 8);
        AZ_CSB;
        tap = ntap; cb = ncbi;
    }
#undef AZ_CDMA_W
#undef AZ_CDMA_IMG
#undef AZ_CLDA
#undef AZ_CLDB
#undef AZ_CMMA
#undef AZ_CSB
#undef AZ_CMIX
    // epilogue through LDS in two halves of 128 rows (the buffers are dead; [128][128] bf16 with a 272-byte row stride = 34 KiB):
    // whole 256-byte row segments, 16 bytes per lane, instead of 8-byte stores in 32-byte pieces of 16 different rows
    constexpr int EP_STRIDE = 272;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wr == half) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int nl = wc * 64 + nt * 16 + fq * 4;
                const float4 bv = *(const float4*)(d.bias + n0 + nl);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const int rl = mt * 16 + frow;                       // row inside the half
                    float r0 = acc[mt][nt][0] + bv.x, r1 = acc[mt][nt][1] + bv.y, r2 = acc[mt][nt][2] + bv.z,
                          r3 = acc[mt][nt][3] + bv.w;
                    if (d.relu) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); r2 = fmaxf(r2, 0.f); r3 = fmaxf(r3, 0.f); }
                    uint2 o;
                    o.x = pack_bf16x2(r0, r1);
                    o.y = pack_bf16x2(r2, r3);
                    *(uint2*)(smem + rl * EP_STRIDE + nl * 2) = o;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 128 * 16 / 256; ++it) {
            const int idx = it * 256 + tid;
            const int rl = idx >> 4, c = idx & 15;
            const int ml = half * 128 + rl;
            const int m = b0 * 42 + ml;
            if (ml >= IMG_ROWS || m >= M) continue;
            *(uint4*)(d.out + (size_t)m * d.N + n0 + c * 8) = *(const uint4*)(smem + rl * EP_STRIDE + c * 16);
        }
        __syncthreads();
    }
}

// ---- 'valid' 3x3 convs (conv3: [6][7][C] -> [4][5][C], conv4: [4][5][C] -> [2][3][C]) image-resident, two 4-wave
// workgroups per CU ---------------------------------------------------------------------------------------------------
// The lockstep argument of k_conv_img2 for the 'valid' convs.  Tile = NB boards (NB*OH*OW output rows, padded to 128 or
// 256) x NCOL channels; every wave owns 128 rows x 64 columns (8 x 4 accumulators, 64 MFMAs per K-step):
//   conv3: NB = 12 (240 of 256 rows), NCOL = 128, waves 2 x 2; LDS = 63 KiB image + 16 KiB weights
//   (conv4: NB = 19 (114 of 128 rows), NCOL = 256, waves 1 x 4, 47.5 KiB image + 32 KiB weights -- measured neutral, not used)
// i.e. 80 KiB = the boards' input image (one buffer) + ONE weight buffer: both 32-deep halves of the step's weight
// fragments are read into registers first, a barrier behind those reads frees the buffer, and the next weight tile is
// DMA'd under the rest of the step.  A 'valid' conv needs no padding logic: output (y, x) of a board reads image row
// (y+ky)*IW + (x+kx).  Same K order: bit-identical.
// The PLANES layout of the conv3 LDS image (conv_valid_tile<.., true>), see Conv3Tables
constexpr int C3_PLANE_BYTES = 32768;         // one chunk-parity plane: 512 cells of 64 B
constexpr int C3_TAB_INV = 512;               // uint16 per LDS row cell: source image row | swizzle << 10
constexpr int C3_TAB_RD = 9 * 2 * 16 * 8;     // uint16 per (tap, wave row, fragment row, row tile): the cell's byte offset in its plane
constexpr int C3_NB = 12;     // (conv4 as <L, 19, 4, 5, 4> is bit-identical too and was measured neutral: it stays on k_gemm256)

// ---- the same tile with the LDS-DMA issued from inline asm and a software-pipelined K-step ---------------------------
// hipcc models `__builtin_amdgcn_global_load_lds` as a FLAT access that may touch LDS and global memory at once: while one
// is pending EVERY s_waitcnt it inserts is vmcnt(0) / lgkmcnt(0), so in k_conv_valid_img2 each MFMA cluster waits for the
// fragment reads issued just before it (meant for the NEXT cluster) -- three exposed LDS latencies per K-step.  Issued
// from inline asm (saddr form: uniform base in SGPRs + one loop-invariant 32-bit offset VGPR per piece, LDS base in M0)
// the compiler does not see the DMA, its own lgkmcnt waits become counted, and the waits for the DMA are the explicit
// vmcnt(0) + barrier pairs below.  K-step schedule (fbX / faX of step k were requested under step k-1's last cluster):
//   reads fbY, faY(ks0) | MFMA fbX x faX(ks0) | lgkmcnt(0), barrier: W(k) is in registers everywhere | DMA W(k+1)
//   reads faX(ks1) | MFMA fbX x faY(ks0) | reads faY(ks1) | MFMA fbY x faX(ks1) | vmcnt(0), barrier: W(k+1) landed
//   reads fbX, faX(ks0) of step k+1 | MFMA fbY x faY(ks1)
// Same K order per accumulator as k_conv_valid_img2 (and every other conv3 kernel): bit-identical.

template <int LAYER, int NB, int IH, int IW, bool STAMP = false, int ABLATE = 0, bool IL = false>   // IL: fragment reads interleaved into the MFMA clusters (sched_group_barrier); STAMP: diagnostic build, per-segment s_memtime sums of wave 0 into d.dbg; ABLATE (timing only, WRONG results): 1 no image switch, 2 + no wait for the weight DMA
__global__ __launch_bounds__(256, 2) void k_conv_valid_pipe(const GemmDesc d) {
    constexpr int OH = IH - 2, OW = IW - 2, OUT_PER = OH * OW, IN_PER = IH * IW;
    constexpr int OUT_ROWS = NB * OUT_PER, IMG_R = NB * IN_PER;
    constexpr int NCOL = 128;
    constexpr int IMG_BYTES = (IMG_R * 128 + 1023) / 1024 * 1024;
    constexpr int IPIECES = (IMG_R + 31) / 32, WPIECES = NCOL / 32;       // 1 KiB DMA pieces per wave
    static_assert(IMG_BYTES + NCOL * 128 <= 81920, "two workgroups must fit a CU's 160 KiB");
    static_assert(IMG_R % 8 == 0 && NB <= 16, "whole 8-row DMA sub-pieces; NetWorkspace keeps 16 boards of slack");
    __shared__ __attribute__((aligned(16))) unsigned char smem[IMG_BYTES + NCOL * 128];   // img | w
    const int n_boards = (int)(*d.n_dev);
    const int M = n_boards * OUT_PER;
    if (M <= d.m_min) return;                          // the small-batch kernel launched beside this one takes the batch
    const int C = d.cin;
    const int NT = d.N / NCOL;
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int ntile = j % NT, mtile = (j / NT) * 8 + xcd;
    const int b0 = mtile * NB, n0 = ntile * NCOL;
    if (b0 >= n_boards) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ lrow;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // DMA addresses: one loop-invariant 32-bit lane offset per operand; piece q adds a uniform stride to the SGPR base.  Image rows
    // past the batch's last board are read unclamped (the workspace keeps 16 boards of slack; their output rows are never stored).
    const uint32_t i_ob = (uint32_t)((b0 * IN_PER + wave * 8 + lrow) * C + chunk * 8) * 2u;
    const uint32_t w_ob = (uint32_t)((n0 + wave * 8 + lrow) * d.K + chunk * 8) * 2u;
    const uint32_t lds_img = (uint32_t)(uintptr_t)(lds_ptr)(smem + wave * 1024);
    const uint32_t lds_w = (uint32_t)(uintptr_t)(lds_ptr)(smem + IMG_BYTES + wave * 1024);
    const size_t i_stride = (size_t)64 * C, w_stride = (size_t)64 * d.K;          // 32 rows, in bytes
#define AZ_PDMA_W(kk_)                                                                                       \
    {                                                                                                        \
        const char* wbase = (const char*)(d.W + (kk_));                                                      \
        _Pragma("unroll") for (int q_ = 0; q_ < WPIECES; ++q_) lds_dma16(wbase + q_ * w_stride, w_ob, lds_w + q_ * 4096); \
    }
#define AZ_PDMA_IMG(cb_)                                                                                     \
    {                                                                                                        \
        const char* ibase = (const char*)(d.A + (cb_) * 64);                                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < IPIECES; ++q_)                                               \
            if ((q_ * 4 + 3) * 8 + 7 < IMG_R || (q_ * 4 + wave) * 8 + 7 < IMG_R)                             \
                lds_dma16(ibase + q_ * i_stride, i_ob, lds_img + q_ * 4096);                                 \
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4, fsw = lane & 7;
    int rbase[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        int ml = wr * 128 + mt * 16 + frow;
        ml = ml < OUT_ROWS ? ml : 0;
        const int bl = ml / OUT_PER, p = ml - bl * OUT_PER, y = p / OW, x = p - y * OW;
        rbase[mt] = bl * IN_PER + y * IW + x;
    }
    const int b_row0 = IMG_BYTES + (wc * 64 + frow) * 128;
    const int coffB0 = ((0 + fq) ^ fsw) << 4, coffB1 = ((4 + fq) ^ fsw) << 4;
#define AZ_PLDA(dst_, mt0_, ks_, dt_)                                                                        \
    if constexpr (ABLATE < 5) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                       \
        const int r_ = rbase[(mt0_) + i_] + (dt_);                                                           \
        dst_[i_] = *(const bf16x8*)(smem + r_ * 128 + ((((ks_) * 4 + fq) ^ (r_ & 7)) << 4));                 \
    }
#define AZ_PLDB(dst_, coff_)                                                                                 \
    if constexpr (ABLATE < 5) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                         \
        dst_[i_] = *(const bf16x8*)(smem + b_row0 + i_ * 2048 + (coff_));
#define AZ_PMMA(mt0_, fb_, fa_)                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                         \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                     \
            acc[(mt0_) + i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb_[j_], fa_[i_], acc[(mt0_) + i_][j_], 0, 0, 0);
#define AZ_PSB __builtin_amdgcn_sched_barrier(0)
#define AZ_PFENCE if constexpr (!IL) __builtin_amdgcn_sched_barrier(0)
    // IL: the region's reads (for the NEXT cluster) go between this cluster's MFMAs: rep_ x { nmf_ MFMAs, 1 LDS read }
#define AZ_PMIX(nmf_, rep_, tail_)                                                                           \
    if constexpr (IL) {                                                                                      \
        _Pragma("unroll") for (int g_ = 0; g_ < (rep_); ++g_) {                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, (nmf_), 0);                                          \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                               \
        }                                                                                                    \
        if ((tail_) > 0) __builtin_amdgcn_sched_group_barrier(0x008, (tail_), 0);                            \
    }
    AZ_PDMA_W(0);
    AZ_PDMA_IMG(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8 fbX[4], fbY[4], faX[4], faY[4];
    if constexpr (ABLATE >= 5) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fbX[i] = fbY[i] = faX[i] = faY[i] = *(const bf16x8*)(smem + lane * 16 + i * 1024);
    }
    AZ_PLDB(fbX, coffB0);
    AZ_PLDA(faX, 0, 0, 0);
    const int ncb = C / 64;
    const int nk = ABLATE == -2 ? 0 : ncb * 9;          // ABLATE -2: prologue + epilogue only
    int cb = 0, tap = 0, dt = 0;
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0, t_begin = 0, r_begin = 0;
    if constexpr (STAMP) { t_begin = tprev = __builtin_amdgcn_s_memtime(); r_begin = __builtin_amdgcn_s_memrealtime(); }
#define AZ_PSTAMP(i_) if constexpr (STAMP) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg[i_] += t_ - tprev; tprev = t_; }
    for (int kt = 0; kt < nk; ++kt) {
        const bool sw = tap == 8;
        const int ntap = sw ? 0 : tap + 1, ncbi = sw ? cb + 1 : cb;
        const int nky = ntap / 3, ndt = nky * IW + (ntap - nky * 3);
        const int kk = kt + 1 < nk ? ntap * C + ncbi * 64 : 8 * C + cb * 64;      // last step: re-fetch its own tile (unused)
        AZ_PLDB(fbY, coffB1);
        AZ_PLDA(faY, 4, 0, dt);
        AZ_PFENCE;
        AZ_PMMA(0, fbX, faX);
        AZ_PMIX(1, 8, 8);
        AZ_PSB;
        AZ_PSTAMP(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): this step's weight fragments are in registers
        if constexpr (ABLATE < 4) __builtin_amdgcn_s_barrier();
        AZ_PSB;
        AZ_PSTAMP(1);
        if constexpr (ABLATE < 3) AZ_PDMA_W(kk);
        AZ_PSTAMP(2);
        AZ_PLDA(faX, 0, 1, dt);
        AZ_PFENCE;
        AZ_PMMA(4, fbX, faY);
        AZ_PMIX(2, 4, 8);
        AZ_PSB;
        AZ_PSTAMP(3);
        AZ_PLDA(faY, 4, 1, dt);
        AZ_PFENCE;
        AZ_PMMA(0, fbY, faX);
        AZ_PMIX(2, 4, 8);
        AZ_PSB;
        AZ_PSTAMP(4);
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // my reads of the image slice are done (they are: one cluster old)
        if constexpr (ABLATE < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the next weight tile has landed
        if constexpr (ABLATE < 4) __builtin_amdgcn_s_barrier();
        AZ_PSB;
        AZ_PSTAMP(5);
        if (ABLATE <= 0 && sw && ncbi < ncb) {                                  // single image buffer: the switch is covered by the CU's other workgroup
            AZ_PDMA_IMG(ncbi);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        AZ_PSB;
        AZ_PSTAMP(6);
        AZ_PLDB(fbX, coffB0);                                    // next step's first fragments, under this step's last cluster
        AZ_PLDA(faX, 0, 0, ndt);
        AZ_PFENCE;
        AZ_PMMA(4, fbY, faY);
        AZ_PMIX(1, 8, 8);
        AZ_PSB;
        AZ_PSTAMP(7);
        tap = ntap; cb = ncbi; dt = ndt;
    }
    if constexpr (STAMP) {
        if (tid == 0 && d.dbg && blockIdx.x < 128) {
            for (int i = 0; i < 8; ++i) d.dbg[16 * blockIdx.x + i] = seg[i];
            d.dbg[16 * blockIdx.x + 8] = __builtin_amdgcn_s_memtime() - t_begin;
            d.dbg[16 * blockIdx.x + 9] = __builtin_amdgcn_s_memrealtime() - r_begin;
        }
    }
#undef AZ_PSTAMP
#undef AZ_PDMA_W
#undef AZ_PDMA_IMG
#undef AZ_PLDA
#undef AZ_PLDB
#undef AZ_PMMA
#undef AZ_PSB
#undef AZ_PFENCE
#undef AZ_PMIX
    // Epilogue through LDS: a lane holds 4 consecutive channels of 32 (row tile, column tile) pairs -- stored directly that is 32
    // eight-byte stores per lane in 32-byte pieces of 16 different rows each (7 % of the kernel at 3072 rows).  The image and weight
    // buffers are dead now: the tile (+ bias, ReLU, bf16) goes to LDS as [240 rows][128 channels] with a 272-byte row stride
    // (conflict-free for both directions), and leaves as whole 256-byte row segments, 16 bytes per lane.
    constexpr int EP_STRIDE = NCOL * 2 + 16;
    static_assert(OUT_ROWS * EP_STRIDE <= IMG_BYTES + NCOL * 128, "the output tile must fit the dead buffers");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                     // every wave is past its last fragment read (and the unused last DMA has landed)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int nl = wc * 64 + nt * 16 + fq * 4;
        const float4 bv = *(const float4*)(d.bias + n0 + nl);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int ml = wr * 128 + mt * 16 + frow;
            if (ml >= OUT_ROWS) continue;
            float r0 = acc[mt][nt][0] + bv.x, r1 = acc[mt][nt][1] + bv.y, r2 = acc[mt][nt][2] + bv.z,
                  r3 = acc[mt][nt][3] + bv.w;
            if (d.relu) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); r2 = fmaxf(r2, 0.f); r3 = fmaxf(r3, 0.f); }
            uint2 o;
            o.x = pack_bf16x2(r0, r1);
            o.y = pack_bf16x2(r2, r3);
            *(uint2*)(smem + ml * EP_STRIDE + nl * 2) = o;
        }
    }
    __syncthreads();
    constexpr int EP_CHUNKS = OUT_ROWS * (NCOL / 8);        // 16-byte chunks of the tile
#pragma unroll
    for (int it = 0; it < (EP_CHUNKS + 255) / 256; ++it) {
        const int idx = it * 256 + tid;
        const int ml = idx / (NCOL / 8), c = idx - ml * (NCOL / 8);
        const int m = b0 * OUT_PER + ml;
        if (idx >= EP_CHUNKS || m >= M) continue;
        const uint4 v = *(const uint4*)(smem + ml * EP_STRIDE + c * 16);
        if (ABLATE >= 0 || v.x == 0x12345678u) *(uint4*)(d.out + (size_t)m * d.N + n0 + c * 8) = v;      // ABLATE -1: the kernel without its stores
    }
}

// ---- the image-resident 'valid' conv tile, generalised: per-wave tile = MT x NTW accumulators of 16x16 ------------------------------
// Waves 2 x 2; a wave owns 16*MT rows x 16*NTW columns, the workgroup 32*MT rows (NB boards' outputs) x 32*NTW columns.  The pipeline
// is k_conv_valid_pipe's (asm-issued LDS-DMA, one weight buffer, fragment reads of the next cluster between this cluster's MFMAs):
// four clusters of (MT/2) x NTW MFMAs per K-step.  Same K order per accumulator: bit-identical to every other kernel of the layer.
//   conv3 full tile   NB 12, 6x7, MT 8, NTW 4   240 of 256 rows x 128 channels, 63 + 16 KiB (= k_conv_valid_pipe)
//   conv3 half tile   NB 12, 6x7, MT 8, NTW 2   the same boards x 64 channels, 63 + 8 KiB: two of them share a CU where a full tile would
//                                               run alone (k_conv3_auto)
//   (conv4 as NB 20, 4x5, MT 4, NTW 4 -- 120 of 128 rows x 128 channels, 50 + 16 KiB -- was built and measured in round 3: bit-identical and
//    slower than the ring at every batch size, 64 against 59 us at 2300 rows, 258 against 197 at 8192: profiles/README.md; removed)
template <int NB, int IH, int IW, int MT, int NTW, bool PLANES = false>
struct ConvTile {
    static constexpr int OH = IH - 2, OW = IW - 2, OUT_PER = OH * OW, IN_PER = IH * IW;
    static constexpr int OUT_ROWS = NB * OUT_PER, IMG_R = NB * IN_PER;
    static constexpr int NCOL = 32 * NTW, WROWS = 16 * MT;
    static constexpr int IMG_BYTES = PLANES ? 2 * C3_PLANE_BYTES : (IMG_R * 128 + 1023) / 1024 * 1024;
    static constexpr int LDS_BYTES = IMG_BYTES + NCOL * 128;
    static constexpr int EP_STRIDE = NCOL * 2 + 16;
    static_assert(OUT_ROWS <= 2 * WROWS && IMG_R % 8 == 0 && MT % 2 == 0, "tile");
    static_assert(LDS_BYTES <= 81920, "two workgroups must fit a CU's 160 KiB");
    static_assert(OUT_ROWS * EP_STRIDE <= LDS_BYTES, "the output tile must fit the dead buffers");
};

template <int NB, int IH, int IW, int MT, int NTW, bool PLANES = false>
__device__ __forceinline__ void conv_valid_tile(const GemmDesc& d, unsigned char* smem, const int b0 /*first board*/, const int n0 /*first column*/,
                                                const int n_boards) {
    using T = ConvTile<NB, IH, IW, MT, NTW, PLANES>;
    static_assert(!PLANES || (NB == C3_NB && IH == 6 && IW == 7 && MT == 8), "the cell maps are conv3's");
    constexpr int OUT_PER = T::OUT_PER, IN_PER = T::IN_PER, OUT_ROWS = T::OUT_ROWS, IMG_R = T::IMG_R, NCOL = T::NCOL, OW = T::OW;
    constexpr int IMG_BYTES = T::IMG_BYTES;
    constexpr int IPIECES = (IMG_R + 31) / 32, WPIECES = NCOL / 32;       // 1 KiB DMA pieces per wave
    constexpr int MH = MT / 2;                                           // row tiles per cluster
    constexpr int NM = MH * NTW;                                         // MFMAs per cluster
    const int M = n_boards * OUT_PER;
    const int C = d.cin;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ lrow;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // DMA addresses: one loop-invariant 32-bit lane offset per operand; piece q adds a uniform stride to the SGPR base.  Image rows past
    // the batch's last board are read unclamped (the workspace keeps a tile of slack; their output rows are never stored).
    const uint32_t i_ob = (uint32_t)((b0 * IN_PER + wave * 8 + lrow) * C + chunk * 8) * 2u;
    // PLANES: LDS cell n = q * 256 + wave * 64 + lane of piece q is chunk position lane & 3 of LDS row (q & 7) * 64 + wave * 16 + (lane >> 2)
    // in plane q >> 3; the row's source (image row, swizzle) comes from the table, the plane is the chunk's low bit (16 bytes, uniform)
    uint32_t i_obp[8];
    if constexpr (PLANES) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t e = d.c3tab[j * 64 + wave * 16 + (lane >> 2)];
            i_obp[j] = (uint32_t)((b0 * IN_PER + (int)(e & 1023u)) * C + (int)((((uint32_t)lane & 3u) ^ ((e >> 10) & 3u)) << 4)) * 2u;
        }
    }
    const uint32_t w_ob = (uint32_t)((n0 + wave * 8 + lrow) * d.K + chunk * 8) * 2u;
    const uint32_t lds_img = (uint32_t)(uintptr_t)(lds_ptr)(smem + wave * 1024);
    const uint32_t lds_w = (uint32_t)(uintptr_t)(lds_ptr)(smem + IMG_BYTES + wave * 1024);
    const size_t i_stride = (size_t)64 * C, w_stride = (size_t)64 * d.K;          // 32 rows, in bytes
#define AZ_TDMA_W(kk_)                                                                                       \
    {                                                                                                        \
        const char* wbase = (const char*)(d.W + (kk_));                                                      \
        _Pragma("unroll") for (int q_ = 0; q_ < WPIECES; ++q_) lds_dma16(wbase + q_ * w_stride, w_ob, lds_w + q_ * 4096); \
    }
#define AZ_TDMA_IMG(cb_)                                                                                     \
    {                                                                                                        \
        const char* ibase = (const char*)(d.A + (cb_) * 64);                                                 \
        if constexpr (PLANES) {                                                                              \
            _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) lds_dma16(ibase + (q_ >> 3) * 16, i_obp[q_ & 7], lds_img + q_ * 4096); \
        } else {                                                                                             \
            _Pragma("unroll") for (int q_ = 0; q_ < IPIECES; ++q_)                                           \
                if ((q_ * 4 + 3) * 8 + 7 < IMG_R || (q_ * 4 + wave) * 8 + 7 < IMG_R)                         \
                    lds_dma16(ibase + q_ * i_stride, i_ob, lds_img + q_ * 4096);                             \
        }                                                                                                    \
    }
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < NTW; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4, fsw = lane & 7;
    int rbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int ml = wr * T::WROWS + mt * 16 + frow;
        ml = ml < OUT_ROWS ? ml : 0;
        const int bl = ml / OUT_PER, p = ml - bl * OUT_PER, y = p / OW, x = p - y * OW;
        rbase[mt] = bl * IN_PER + y * IW + x;
    }
    const int b_row0 = IMG_BYTES + (wc * (16 * NTW) + frow) * 128;
    const int coffB0 = ((0 + fq) ^ fsw) << 4, coffB1 = ((4 + fq) ^ fsw) << 4;
    // PLANES: the 8 row tiles' cells of one tap are one 16-byte table entry per lane (tq: this step's tap, ntq: the next step's, loaded a
    // step ahead); chunk c = ks * 4 + fq of a cell: plane c & 1, position (c >> 1) ^ swizzle (the swizzle is folded into the table value)
    const uint4* rdtab = (const uint4*)(d.c3tab + C3_TAB_INV) + (wr * 16 + frow);
    const uint32_t lane_x = (((uint32_t)fq & 1u) << 15) | (((uint32_t)fq >> 1) << 4);
    uint32_t tq[4] = {0, 0, 0, 0}, ntq[4] = {0, 0, 0, 0};
    if constexpr (PLANES) { const uint4 v = rdtab[0]; tq[0] = v.x; tq[1] = v.y; tq[2] = v.z; tq[3] = v.w; }
#define AZ_TCELL(q_, mt_) ((((mt_) & 1) ? q_[(mt_) >> 1] >> 16 : q_[(mt_) >> 1] & 0xFFFFu) ^ lane_x)
#define AZ_TLDA(dst_, mt0_, ks_, dt_, q_)                                                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < MH; ++i_) {                                                      \
        if constexpr (PLANES) {                                                                              \
            dst_[i_] = *(const bf16x8*)(smem + (AZ_TCELL(q_, (mt0_) + i_) ^ ((ks_) << 5)));                  \
        } else {                                                                                             \
            const int r_ = rbase[(mt0_) + i_] + (dt_);                                                       \
            dst_[i_] = *(const bf16x8*)(smem + r_ * 128 + ((((ks_) * 4 + fq) ^ (r_ & 7)) << 4));             \
        }                                                                                                    \
    }
#define AZ_TLDB(dst_, coff_)                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < NTW; ++i_)                                                       \
        dst_[i_] = *(const bf16x8*)(smem + b_row0 + i_ * 2048 + (coff_));
#define AZ_TMMA(mt0_, fb_, fa_)                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < MH; ++i_)                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < NTW; ++j_)                                                   \
            acc[(mt0_) + i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb_[j_], fa_[i_], acc[(mt0_) + i_][j_], 0, 0, 0);
#define AZ_TSB __builtin_amdgcn_sched_barrier(0)
    // the region's nr_ reads (for the NEXT cluster) go between this cluster's NM MFMAs: nr_ x { nmf MFMAs, 1 LDS read }, then the rest
#define AZ_TMIX(nr_)                                                                                         \
    {                                                                                                        \
        constexpr int nmf_ = NM / (2 * (nr_)) > 1 ? NM / (2 * (nr_)) : 1;                                    \
        _Pragma("unroll") for (int g_ = 0; g_ < (nr_); ++g_) {                                               \
            __builtin_amdgcn_sched_group_barrier(0x008, nmf_, 0);                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                               \
        }                                                                                                    \
        if (NM - nmf_ * (nr_) > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - nmf_ * (nr_), 0);        \
    }
    static_assert(NM >= NTW + MH, "a cluster must hold at least one MFMA per read of its region");
    AZ_TDMA_W(0);
    AZ_TDMA_IMG(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8 fbX[NTW], fbY[NTW], faX[MH], faY[MH];
    AZ_TLDB(fbX, coffB0);
    AZ_TLDA(faX, 0, 0, 0, tq);
    const int ncb = C / 64;
    const int nk = ncb * 9;
    int cb = 0, tap = 0, dt = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool sw = tap == 8;
        const int ntap = sw ? 0 : tap + 1, ncbi = sw ? cb + 1 : cb;
        const int nky = ntap / 3, ndt = nky * IW + (ntap - nky * 3);
        const int kk = kt + 1 < nk ? ntap * C + ncbi * 64 : 8 * C + cb * 64;      // last step: re-fetch its own tile (unused)
        if constexpr (PLANES) { const uint4 v = rdtab[ntap * 32]; ntq[0] = v.x; ntq[1] = v.y; ntq[2] = v.z; ntq[3] = v.w; }
        AZ_TLDB(fbY, coffB1);
        AZ_TLDA(faY, MH, 0, dt, tq);
        AZ_TMMA(0, fbX, faX);
        AZ_TMIX(NTW + MH);
        AZ_TSB;
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): this step's weight fragments are in registers
        __builtin_amdgcn_s_barrier();
        AZ_TSB;
        AZ_TDMA_W(kk);
        AZ_TLDA(faX, 0, 1, dt, tq);
        AZ_TMMA(MH, fbX, faY);
        AZ_TMIX(MH);
        AZ_TSB;
        AZ_TLDA(faY, MH, 1, dt, tq);
        AZ_TMMA(0, fbY, faX);
        AZ_TMIX(MH);
        AZ_TSB;
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // my reads of the image slice are done (they are: one cluster old)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the next weight tile has landed
        __builtin_amdgcn_s_barrier();
        AZ_TSB;
        if (sw && ncbi < ncb) {                                  // single image buffer: the switch is covered by the CU's other workgroup
            AZ_TDMA_IMG(ncbi);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        AZ_TSB;
        AZ_TLDB(fbX, coffB0);                                    // next step's first fragments, under this step's last cluster
        AZ_TLDA(faX, 0, 0, ndt, ntq);
        AZ_TMMA(MH, fbY, faY);
        AZ_TMIX(NTW + MH);
        AZ_TSB;
        tap = ntap; cb = ncbi; dt = ndt;
        if constexpr (PLANES) { tq[0] = ntq[0]; tq[1] = ntq[1]; tq[2] = ntq[2]; tq[3] = ntq[3]; }
    }
#undef AZ_TDMA_W
#undef AZ_TDMA_IMG
#undef AZ_TLDA
#undef AZ_TCELL
#undef AZ_TLDB
#undef AZ_TMMA
#undef AZ_TSB
#undef AZ_TMIX
    // epilogue through LDS (the image and weight buffers are dead): [OUT_ROWS][NCOL] bf16 with a padded row stride, out as whole row
    // segments of 16 bytes per lane
    constexpr int EP_STRIDE = T::EP_STRIDE;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                     // every wave is past its last fragment read (and the unused last DMA has landed)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int nl = wc * (16 * NTW) + nt * 16 + fq * 4;
        const float4 bv = *(const float4*)(d.bias + n0 + nl);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int ml = wr * T::WROWS + mt * 16 + frow;
            if (ml >= OUT_ROWS) continue;
            float r0 = acc[mt][nt][0] + bv.x, r1 = acc[mt][nt][1] + bv.y, r2 = acc[mt][nt][2] + bv.z,
                  r3 = acc[mt][nt][3] + bv.w;
            if (d.relu) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); r2 = fmaxf(r2, 0.f); r3 = fmaxf(r3, 0.f); }
            uint2 o;
            o.x = pack_bf16x2(r0, r1);
            o.y = pack_bf16x2(r2, r3);
            *(uint2*)(smem + ml * EP_STRIDE + nl * 2) = o;
        }
    }
    __syncthreads();
    constexpr int EP_CHUNKS = OUT_ROWS * (NCOL / 8);        // 16-byte chunks of the tile
#pragma unroll
    for (int it = 0; it < (EP_CHUNKS + 255) / 256; ++it) {
        const int idx = it * 256 + tid;
        const int ml = idx / (NCOL / 8), c = idx - ml * (NCOL / 8);
        const int m = b0 * OUT_PER + ml;
        if (idx >= EP_CHUNKS || m >= M) continue;
        *(uint4*)(d.out + (size_t)m * d.N + n0 + c * 8) = *(const uint4*)(smem + ml * EP_STRIDE + c * 16);
    }
}

// conv3 with a HALF-TILE TAIL.  A launch's workgroups arrive in rounds of 512 (two per CU); a short last round keeps a few CUs busy for a
// whole workgroup's duration while the rest of the chip idles.  The tiles of such a round are cut in two along the CHANNELS (12 boards x
// 64 channels: the same image, half the weights and half the MFMAs per step), so the tail lasts about 0.6 of a full tile.  The role of a
// workgroup follows from the batch's row count, read on the device; the host launches the full grid for its upper bound plus C3_TAIL
// extra workgroups (the second halves), which exit at once when no tile is cut.
// Measured (tools/layer_times.py, round 3): cutting pays when the halves still run alone on their CUs, i.e. for a last round of at most
// 128 workgroups (3100 rows: 254 -> 240 us; 700 rows: 77 -> 71); a last round of 129-256 cut into 258-512 halves runs two halves per CU
// and is slower than the uncut round (2300 rows: 160 -> 170 us) -- a workgroup alone on its CU is not the 76-us level round 2's notes
// priced it at, so those rounds stay uncut.
constexpr int C3_TAIL = 128;
template <int LAYER, bool PLANES>
__global__ __launch_bounds__(256, 2) void k_conv3_auto(const GemmDesc d, const int full_grid, const int tail_wgs /* C3_TAIL, or 0: no tile is cut */) {
    using TF = ConvTile<C3_NB, 6, 7, 8, 4, PLANES>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TF::LDS_BYTES];
    const int n_boards = (int)(*d.n_dev);
    if (n_boards * d.rows_per_sample <= d.m_min) return;      // the small-batch kernel launched beside this one takes the batch
    if (blockIdx.x == 0 && threadIdx.x == 0 && d.acct && n_boards > 0) { atomicAdd(&d.acct[0], (unsigned long long)n_boards); atomicAdd(&d.acct[1], 1ull); }
    const int NT = d.N / 128;
    const int tiles8 = ((n_boards + C3_NB - 1) / C3_NB + 7) / 8 * 8;
    const int wid = tiles8 * NT;                       // workgroup ids that map to a tile of this batch (the last group of 8 row tiles may be partly empty)
    const int full_rounds = wid / 512 * 512, rem = wid - full_rounds;
    const bool cut = rem > 0 && rem <= tail_wgs;       // the last round would run one workgroup per CU
    int id = blockIdx.x, half = -1;
    if (id >= full_grid) {                             // an extra workgroup: the second half of a cut tile
        if (!cut || id - full_grid >= rem) return;
        id = full_rounds + (id - full_grid);
        half = 1;
    } else if (cut && id >= full_rounds) {
        half = 0;
    }
    const int xcd = id & 7, j = id >> 3;
    const int ntile = j % NT, mtile = (j / NT) * 8 + xcd;
    const int b0 = mtile * C3_NB, n0 = ntile * 128;
    if (b0 >= n_boards) return;
    if (half < 0) conv_valid_tile<C3_NB, 6, 7, 8, 4, PLANES>(d, smem, b0, n0, n_boards);
    else conv_valid_tile<C3_NB, 6, 7, 8, 2, PLANES>(d, smem, b0, n0 + half * 64, n_boards);
}

#ifdef AZ_DIAG
// ---- conv3 as a PING-PONG kernel: ONE 8-wave workgroup per CU, the two waves of every SIMD in opposite phases ------------------------
// k_conv3_auto's two independent 4-wave workgroups per CU interleave fragment reads, DMA issue and MFMAs in every wave's in-order
// stream: the matrix pipe waits whenever both waves of a SIMD wait (SQ_WAIT_INST_ANY 56 % of wave cycles, profiles/r03f_pmc_sq_*).
// Here the two waves of a SIMD belong to ONE workgroup and alternate roles, separated by workgroup barriers: in every slot one of them
// issues nothing but 32 MFMAs on fragments it already holds in registers (the pipe runs back to back), the other one reads the 12
// fragments of ITS next 32 MFMAs and issues the slot's LDS-DMA.  Group 0 = waves 0-3 (rows 0..127 of the tile), group 1 = waves 4-7 (rows
// 128..255), group 1 one barrier behind.
//   tile     12 boards (240 of 256 rows) x 256 channels; wave = 128 rows x 64 channels (8 x 4 accumulators, as in every conv3 kernel)
//   stage s  = (channel block cb, tap, k half): 32 deep.  Weights: 256 channels x 32 k = 16 KiB per stage, TWO stage buffers, streamed from
//            the model's packed copy (ConvNet::wp3: the LDS image of every stage, swizzle included, contiguous -- one DMA piece is 1 KiB
//            of consecutive global bytes).  Image: the boards' 64-channel slice of act2 (63 KiB), TWO buffers: slice cb + 1 streams in
//            during the 36 slots of slice cb, so there is no image switch to wait for.  2 x 63 + 2 x 16 = 158 KiB.
//   slot 2s     group 0: LOAD(s): 12 ds_read_b128, DMA weights of stage s + 1 | group 1: MFMAs of stage s - 1
//   slot 2s + 1 group 0: MFMAs of stage s, then vmcnt(0)                     | group 1: LOAD(s), DMA one piece of image slice cb + 1
// Same K order per accumulator as every other conv3 kernel (channel block outer, tap inner, k 0..31 then 32..63): bit-identical.
constexpr int PP_NB = C3_NB;                       // boards per tile (the cell maps are Conv3Tables')
constexpr int PP_NCOL = 256;                       // channels per tile
constexpr int PP_WSTAGE = PP_NCOL * 64;            // one weight stage: 64-byte rows, chunk q of row n at slot q ^ pp_wperm(n)
// LDS: [chunk-parity plane][image buffer][32 KiB of 64-byte cells] | w[2].  The buffer is bit 15 of an image address (an immediate of
// the ds_read), the plane bit 16 (per lane).
constexpr int PP_LDS = 4 * C3_PLANE_BYTES + 2 * PP_WSTAGE;
constexpr int PP_EP_STRIDE = PP_NCOL * 2 + 16;
static_assert(PP_LDS <= 163840 && PP_NB * 20 * PP_EP_STRIDE <= PP_LDS, "one workgroup per CU: 160 KiB");
AZ_HD int pp_wperm(int n) { return (0x1320 >> (((n >> 2) & 3) * 4)) & 3; }      // {0, 2, 3, 1}[(n >> 2) & 3]: conflict-free ds_read_b128 of 64-byte rows

template <int LAYER, int ABL = 0, int SCHED = 2, int TAIL = 2>     // TAIL: row tiles (x 4 MFMAs) of a stage issued behind its closing barrier; SCHED: barriers per stage (2: a LOAD and a COMPUTE slot; 1: see AZ_QSTAGE); ABL (diagnostic library, timing only, WRONG results): 1 no weight DMA, 2 no image DMA, 4 no fragment reads, 8 no MFMAs, 16 THREE weight buffers (the first overlaps the image: what a third buffer would buy)
__global__ __launch_bounds__(512, 2) void k_conv3_pp(const GemmDesc d) {
    constexpr int OUT_PER = 20, IN_PER = 42, OUT_ROWS = PP_NB * OUT_PER;
    __shared__ __attribute__((aligned(16))) unsigned char smem[PP_LDS];
    const int n_boards = (int)(*d.n_dev);
    const int M = n_boards * OUT_PER;
    if (M <= d.m_min) return;                          // the small-batch kernel launched beside this one takes the batch
    if (blockIdx.x == 0 && threadIdx.x == 0 && d.acct && n_boards > 0) { atomicAdd(&d.acct[0], (unsigned long long)n_boards); atomicAdd(&d.acct[1], 1ull); }
    const int C = d.cin;
    const int NT = d.N / PP_NCOL;
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int ntile = j % NT, mtile = (j / NT) * 8 + xcd;
    const int b0 = mtile * PP_NB, n0 = ntile * PP_NCOL;
    if (b0 >= n_boards) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wr = (wave >> 1) & 1, wc = g * 2 + (wave & 1);   // group (= column half of the tile: each group streams ITS 128 channels' weights), row half, column quarter
    const int w4 = wave & 3;                           // wave within its group
    const int frow = lane & 15, fq = lane >> 4;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr)smem;
    const int ncb = C / 64, nst = ncb * 18;
    const char* wp = (const char*)d.Wp + (size_t)ntile * nst * PP_WSTAGE;
    const uint32_t w_vo = (uint32_t)(wave * 2048 + lane * 16);             // wave w: pieces 2 w, 2 w + 1 of a stage (its own group's half)
    constexpr int R3 = (ABL & 16) ? 1 : 0;             // weight ring of 3 (ablation) or 2 stages
    constexpr int W_BASE = 4 * C3_PLANE_BYTES - R3 * PP_WSTAGE;
    const uint32_t lds_w = lds_base + W_BASE + wave * 2048;
    // image DMA (Conv3Tables' inverse map): piece q of a group's wave w4 fills cells (q & 7) * 64 + w4 * 16 + (lane >> 2) of plane q >> 3;
    // group 0 brings plane 0 (pieces 0..7), group 1 plane 1 (pieces 8..15)
    // (kept packed, two per register: image row | chunk position << 10; the byte offset is rebuilt where a piece is issued)
    uint32_t i_pk[4];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const uint32_t e = d.c3tab[jj * 64 + w4 * 16 + (lane >> 2)];
        const uint32_t pk = (e & 1023u) | ((((uint32_t)lane & 3u) ^ ((e >> 10) & 3u)) << 10);
        if (jj & 1) i_pk[jj >> 1] |= pk << 16; else i_pk[jj >> 1] = pk;
    }
    const uint32_t i_b0 = (uint32_t)(b0 * IN_PER * C) * 2u, i_rs = (uint32_t)C * 2u;
    const uint32_t lds_i = lds_base + w4 * 1024;
#define AZ_QDMA_IMG(q_, cb_)                                                                                                  \
    {                                                                                                                          \
        uint32_t pkr_ = i_pk[((q_) & 7) >> 1];                                                                                 \
        asm volatile("" : "+v"(pkr_));                 /* opaque: the eight offsets are loop-invariant and must not be hoisted (spills) */ \
        const uint32_t pk_ = (((q_) & 1) ? pkr_ >> 16 : pkr_) & 0xFFFFu;                                                       \
        lds_dma16((const char*)(d.A + (cb_) * 64) + ((q_) >> 3) * 16, i_b0 + (pk_ & 1023u) * i_rs + ((pk_ >> 10) << 5),        \
                  lds_i + ((q_) >> 3) * (2 * C3_PLANE_BYTES) + ((cb_) & 1) * C3_PLANE_BYTES + ((q_) & 7) * 4096);              \
    }
    // prologue: the first image slice (plane 0 by waves 0-3, plane 1 by waves 4-7) and weight stage 0
#pragma unroll
    for (int i = 0; i < 8; ++i) AZ_QDMA_IMG(g * 8 + i, 0)
#pragma unroll
    for (int i = 0; i < 2; ++i) lds_dma16(wp + i * 1024, w_vo, lds_w + i * 1024);
    if ((SCHED == 1 && g == 1) || R3) {                 // SCHED 1: group 1 issues one stage further ahead (see AZ_QSTAGE)
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(wp + PP_WSTAGE + i * 1024, w_vo, lds_w + PP_WSTAGE + i * 1024);
    }
    if (SCHED == 1 && g == 1 && R3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(wp + 2 * PP_WSTAGE + i * 1024, w_vo, lds_w + 2 * PP_WSTAGE + i * 1024);
    }
    // fragment cells of all nine taps: 8 row tiles x 16 bit per tap and lane (Conv3Tables' read map), the lane's chunk position folded in
    uint32_t tq[9][4];
    {
        const uint4* rdtab = (const uint4*)(d.c3tab + C3_TAB_INV) + (wr * 16 + frow);
        const uint32_t px = ((uint32_t)fq >> 1) << 4, pxx = px | (px << 16);
#pragma unroll
        for (int t = 0; t < 9; ++t) { const uint4 v = rdtab[t * 32]; tq[t][0] = v.x ^ pxx; tq[t][1] = v.y ^ pxx; tq[t][2] = v.z ^ pxx; tq[t][3] = v.w ^ pxx; }
    }
    const uint32_t lx = ((uint32_t)fq & 1u) << 16;     // the lane's plane
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* wb = smem + W_BASE + (wc * 64 + frow) * 64 + ((fq ^ pp_wperm(frow)) << 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8 fb[4], fa[8];
    if constexpr ((ABL & 4) != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = *(const bf16x8*)(smem + 4 * C3_PLANE_BYTES + lane * 16 + i * 1024);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *(const bf16x8*)(smem + lane * 16 + i * 1024);
    }
    if (SCHED == 2 && g == 1) __builtin_amdgcn_s_barrier();       // SCHED 2: group 1 runs one slot behind group 0
    __builtin_amdgcn_sched_barrier(0);
    int s = 0;
    // ABL 128 (diagnostic): per-segment s_memtime sums of waves 0 and 4 -- LOAD until its reads are back | barrier 1 | MFMA issue |
    // wait for the DMA | barrier 2 -- every stamp sits where lgkmcnt is 0 anyway
    unsigned long long seg[5] = {0, 0, 0, 0, 0}, tprev = 0, t_begin = 0;
    if constexpr ((ABL & 128) != 0) { t_begin = tprev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); }
#define AZ_QSTAMP(i_)                                                                                                          \
    if constexpr ((ABL & 128) != 0) {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        unsigned long long t_;                                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                                          \
        seg[i_] += t_ - tprev; tprev = t_;                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
    }
    // SCHED 1: both groups run the same sequence LOAD(0) COMPUTE(0) LOAD(1) COMPUTE(1) ...; group 0's barrier stands behind every COMPUTE, group 1's
    // behind every LOAD, so between two barriers group 0 does LOAD(i), COMPUTE(i) while group 1 does COMPUTE(i - 1), LOAD(i): ONE barrier
    // per stage keeps the two waves of a SIMD in opposite phases.  Each group opens its interval with its DMA (its half of a weight stage,
    // then at most one image piece) and waits for the weight pieces just before its barrier: a LOAD and a COMPUTE later.
    // weight stage st_ -> buffer st_ & 1 (given as a literal); IMG_: this interval also brings one piece of image slice cb + 1
#define AZ_QDMA(st_, wbuf_, IMG_, TAP_)                                                                                        \
    {                                                                                                                          \
        if (!(ABL & 1) && (st_) < nst) {                                                                                       \
            const char* src_ = wp + (size_t)(st_) * PP_WSTAGE;                                                                 \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) lds_dma16(src_ + i_ * 1024, w_vo, lds_w + (wbuf_) * PP_WSTAGE + i_ * 1024); \
        }                                                                                                                      \
        if (!(ABL & 2) && (IMG_) && cb + 1 < ncb) AZ_QDMA_IMG(g * 8 + (TAP_), cb + 1)                                           \
    }
    // wait for the weight pieces of the interval; an image piece issued behind them may stay in flight
#define AZ_QWAITV(IMG_)                                                                                                        \
    {                                                                                                                          \
        if constexpr (R3) {                            /* the stage issued in THIS interval may stay in flight */               \
            if ((IMG_) && !(ABL & 2) && cb + 1 < ncb) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                          \
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                              \
        } else {                                                                                                               \
            if ((IMG_) && !(ABL & 2) && cb + 1 < ncb) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                          \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
        }                                                                                                                      \
    }
    // one stage: BUF_ (image buffer), TAP_, HALF_ are literals, so every LDS address is a register + an immediate
#define AZ_QLOAD(BUF_, TAP_, HALF_)                                                                                            \
        if constexpr (!(ABL & 4)) _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) fb[nt] = *(const bf16x8*)(wb + (R3 ? ((TAP_) * 2 + (HALF_)) % 3 : (HALF_)) * PP_WSTAGE + nt * 1024); \
        if constexpr (!(ABL & 4)) _Pragma("unroll") for (int mt = 0; mt < 8; ++mt) {                                           \
            uint32_t a_;   /* asm volatile: the 144 addresses of a slice are loop-invariant and must NOT be hoisted (spills) */ \
            if (HALF_) asm volatile("v_perm_b32 %0, %1, %2, %3\n\tv_xor_b32 %0, 32, %0" : "=&v"(a_) : "v"(lx), "v"(tq[TAP_][mt >> 1]), "s"((mt & 1) ? 0x0C060302u : 0x0C060100u)); \
            else asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a_) : "v"(lx), "v"(tq[TAP_][mt >> 1]), "s"((mt & 1) ? 0x0C060302u : 0x0C060100u)); \
            fa[mt] = *(const bf16x8*)(smem + (BUF_) * C3_PLANE_BYTES + a_);                                                    \
        }
    // row tiles [mt0_, mt1_) of the stage's 8 x 4 MFMAs
#define AZ_QCOMPUTE(mt0_, mt1_)                                                                                                \
        if constexpr (!(ABL & 96)) __builtin_amdgcn_s_setprio(1);                                                              \
        if constexpr (!(ABL & 8)) _Pragma("unroll") for (int mt = (mt0_); mt < (mt1_); ++mt)                                   \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                                   \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);                   \
        else { _Pragma("unroll") for (int mt = 0; mt < 8; ++mt) asm volatile("" :: "v"(fa[mt])); _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) asm volatile("" :: "v"(fb[nt])); } \
        if constexpr (!(ABL & 96)) __builtin_amdgcn_s_setprio(0);
    // one stage: BUF_ (image buffer), TAP_, HALF_ are literals, so every LDS address is a register + an immediate
#define AZ_QSTAGE(BUF_, TAP_, HALF_)                                                                                           \
    if constexpr (SCHED == 2) {                                                                                                \
        /* two barriers per stage: a LOAD slot and a COMPUTE slot per wave, group 1 one slot behind group 0 */                  \
        AZ_QDMA(s + 1 + R3, R3 ? ((TAP_) * 2 + (HALF_) + 2) % 3 : 1 - (HALF_), (HALF_) == 0 && (TAP_) < 8, TAP_)               \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        AZ_QLOAD(BUF_, TAP_, HALF_)                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        __builtin_amdgcn_s_waitcnt(0xC07F);            /* lgkmcnt(0): my reads are done before anybody may overwrite them */   \
        AZ_QSTAMP(0)                                                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                          \
        AZ_QSTAMP(1)                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        AZ_QCOMPUTE(0, 8 - TAIL)                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        AZ_QSTAMP(2)                                                                                                           \
        AZ_QWAITV((HALF_) == 0 && (TAP_) < 8)                                                                                  \
        AZ_QSTAMP(3)                                                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                          \
        AZ_QSTAMP(4)                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        /* the last TAIL row tiles' MFMAs issue BEHIND the barrier: they keep the matrix pipe busy while the other group starts */ \
        if constexpr (TAIL > 0) { AZ_QCOMPUTE(8 - TAIL, 8) __builtin_amdgcn_sched_barrier(0); }                                \
        ++s;                                                                                                                   \
    } else {                                                                                                                   \
        if (g == 0) AZ_QDMA(s + 1 + R3, R3 ? ((TAP_) * 2 + (HALF_) + 2) % 3 : 1 - (HALF_), (HALF_) == 0 && (TAP_) < 8, TAP_)   \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        AZ_QLOAD(BUF_, TAP_, HALF_)                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        if (g == 1) {                                                                                                          \
            __builtin_amdgcn_s_waitcnt(0xC07F);        /* lgkmcnt(0): my reads are done before anybody may overwrite them */   \
            AZ_QWAITV((HALF_) == 1 && (TAP_) < 8)      /* what I issued behind the previous stage's barrier */                 \
            __builtin_amdgcn_s_barrier();                                                                                      \
            AZ_QDMA(s + 2 + R3, R3 ? ((TAP_) * 2 + (HALF_)) % 3 : (HALF_), (HALF_) == 0 && (TAP_) < 8, TAP_)                   \
        }                                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        AZ_QCOMPUTE(0, 8)                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        if (g == 0) {                                                                                                          \
            AZ_QWAITV((HALF_) == 0 && (TAP_) < 8)                                                                              \
            __builtin_amdgcn_s_barrier();                                                                                      \
        }                                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        ++s;                                                                                                                   \
    }
#define AZ_QSLICE(BUF_)                                                                                                        \
    AZ_QSTAGE(BUF_, 0, 0) AZ_QSTAGE(BUF_, 0, 1) AZ_QSTAGE(BUF_, 1, 0) AZ_QSTAGE(BUF_, 1, 1) AZ_QSTAGE(BUF_, 2, 0) AZ_QSTAGE(BUF_, 2, 1) \
    AZ_QSTAGE(BUF_, 3, 0) AZ_QSTAGE(BUF_, 3, 1) AZ_QSTAGE(BUF_, 4, 0) AZ_QSTAGE(BUF_, 4, 1) AZ_QSTAGE(BUF_, 5, 0) AZ_QSTAGE(BUF_, 5, 1) \
    AZ_QSTAGE(BUF_, 6, 0) AZ_QSTAGE(BUF_, 6, 1) AZ_QSTAGE(BUF_, 7, 0) AZ_QSTAGE(BUF_, 7, 1) AZ_QSTAGE(BUF_, 8, 0) AZ_QSTAGE(BUF_, 8, 1)
    for (int cb = 0; cb < ncb; ++cb) {                 // ncb is even (C % 256 == 0): two slices per trip, the image buffer a literal
        AZ_QSLICE(0)
        ++cb;
        AZ_QSLICE(1)
    }
#undef AZ_QDMA
#undef AZ_QWAITV
#undef AZ_QLOAD
#undef AZ_QCOMPUTE
#undef AZ_QSTAMP
    if constexpr ((ABL & 128) != 0) {
        if ((wave & 3) == 0 && lane == 0 && d.dbg && blockIdx.x < 128) {
            for (int i = 0; i < 5; ++i) d.dbg[16 * blockIdx.x + g * 8 + i] = seg[i];
            d.dbg[16 * blockIdx.x + g * 8 + 7] = __builtin_amdgcn_s_memtime() - t_begin;
        }
    }
    if (SCHED == 2 && g == 0) __builtin_amdgcn_s_barrier();       // pairs with group 1's last COMPUTE barrier
#undef AZ_QSLICE
#undef AZ_QSTAGE
#undef AZ_QDMA_IMG
    __builtin_amdgcn_sched_barrier(0);
    // epilogue through LDS (every buffer is dead): [240][256] bf16 with a 528-byte row stride, out as whole 512-byte row segments
    // (the lane constants are rebuilt from an opaque thread id: computed before the K loop they would be spilled across it)
    __syncthreads();
    int etid = (int)threadIdx.x;
    asm volatile("" : "+v"(etid));
    const int efrow = etid & 15, efq = (etid >> 4) & 3;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int nl = wc * 64 + nt * 16 + efq * 4;
        const float4 bv = *(const float4*)(d.bias + n0 + nl);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int ml = wr * 128 + mt * 16 + efrow;
            if (ml >= OUT_ROWS) continue;
            float r0 = acc[mt][nt][0] + bv.x, r1 = acc[mt][nt][1] + bv.y, r2 = acc[mt][nt][2] + bv.z, r3 = acc[mt][nt][3] + bv.w;
            if (d.relu) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); r2 = fmaxf(r2, 0.f); r3 = fmaxf(r3, 0.f); }
            uint2 o;
            o.x = pack_bf16x2(r0, r1);
            o.y = pack_bf16x2(r2, r3);
            *(uint2*)(smem + ml * PP_EP_STRIDE + nl * 2) = o;
        }
    }
    __syncthreads();
    constexpr int EP_CHUNKS = OUT_ROWS * (PP_NCOL / 8);
#pragma unroll
    for (int it = 0; it < (EP_CHUNKS + 511) / 512; ++it) {
        const int idx = it * 512 + etid;
        const int ml = idx / (PP_NCOL / 8), c = idx - ml * (PP_NCOL / 8);
        const int m = b0 * OUT_PER + ml;
        if (idx >= EP_CHUNKS || m >= M) continue;
        *(uint4*)(d.out + (size_t)m * d.N + n0 + c * 8) = *(const uint4*)(smem + ml * PP_EP_STRIDE + c * 16);
    }
}

#endif   // AZ_DIAG (k_conv3_pp)

// ---- SKINNY GEMM for small batches: operands straight into registers, no LDS, no barrier ---------------------------------------------
// The arena (temp 0: ~20 executed rows per step and model), the drain of a self-play call and single-tree calls run the forward on a
// few dozen rows.  The tiled kernels are then a chain of K / 64 dependent steps of { DMA, wait, barrier, fragment reads, MFMAs } run by
// a handful of workgroups: conv3 took 32 us, conv4 32, fc1 21 whatever the rows (32 x 32 tiles on an 8-stage ring, built first in
// round 3, still took 20 us per layer: the step is latency, not bytes).  Here a WAVE is the unit: it owns MR x NR accumulators of
// 16 x 16 and loads the operand fragments of a K-step directly in the MFMA register layout (lane = row lane & 15, k-group lane >> 4: 16
// contiguous bytes per lane), D K-steps ahead in a register ring; the compiler's counted vmcnt waits retire them; no wave waits for
// another.  A CU takes register loads in at only ~10 B/clk (MI355X_MICROARCH.md; four waves of one CU streaming 64 columns measured
// 36 us for conv3 at ONE row, bound by exactly that), so the point is how many CUs share the weight stream:
//   1 x 1 tiles, D = 8   ceil(M / 16) x N / 16 waves, 2 loads per MFMA: up to 256 waves (conv3 at one row: 64 waves, 13 us)
//   2 x 2 tiles, D = 4   a quarter of the waves, 1 load per MFMA: beyond that
// chosen on the device from the batch's exact row count.  Same K order per accumulator (channel block outer, tap inner; k 0..31 then
// 32..63 of a block) as every other kernel of the layer: bit-identical.
template <int MR, int NR, int D>
__device__ __forceinline__ void gemm_skinny_tile(const GemmDesc& d, const int M, const int mtile, const int ntile, const int lane) {
    const int m0 = mtile * (16 * MR), n0 = ntile * (16 * NR);
    const int frow = lane & 15, fq = lane >> 4;
    const uint16_t* a_ptr[MR];
    const uint16_t* w_ptr[NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        int m = m0 + i * 16 + frow;
        m = m < M ? m : M - 1;
        const int b = m / d.rows_per_sample, r = m - b * d.rows_per_sample;
        const int y = r / d.out_w, x = r - y * d.out_w;
        a_ptr[i] = d.A + (size_t)((b * d.in_h + y) * d.in_w + x) * d.in_c + fq * 8;
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) w_ptr[j] = d.W + (size_t)(n0 + j * 16 + frow) * d.K + fq * 8;
    const int ntaps = d.K / d.cin;
    const int nk = d.K / GBK;
    // K-step walker (channel block outer, tap inner), scalars only
    int ks_tap = 0, ks_kx = 0;
    uint32_t ks_c0 = 0, ks_toff = 0, ks_kk = 0;
    bf16x8 fa0[D][MR], fa1[D][MR], fb0[D][NR], fb1[D][NR];
#define AZ_SLOAD(s_)                                                                                    \
    {                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < MR; ++i_) {                                             \
            fa0[s_][i_] = *(const bf16x8*)(a_ptr[i_] + ks_toff); fa1[s_][i_] = *(const bf16x8*)(a_ptr[i_] + ks_toff + 32); }  \
        _Pragma("unroll") for (int j_ = 0; j_ < NR; ++j_) {                                             \
            fb0[s_][j_] = *(const bf16x8*)(w_ptr[j_] + ks_kk); fb1[s_][j_] = *(const bf16x8*)(w_ptr[j_] + ks_kk + 32); }      \
        ++ks_tap; ++ks_kx; ks_toff += (uint32_t)d.in_c; ks_kk += (uint32_t)d.cin;                       \
        if (ks_kx == d.tap_w) { ks_kx = 0; ks_toff += (uint32_t)((d.in_w - d.tap_w) * d.in_c); }        \
        if (ks_tap == ntaps) { ks_tap = 0; ks_kx = 0; ks_c0 += GBK; ks_toff = ks_c0; ks_kk = ks_c0; }   \
    }
#define AZ_SMMA(s_)                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < MR; ++i_)                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < NR; ++j_) {                                             \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[s_][j_], fa0[s_][i_], acc[i_][j_], 0, 0, 0); \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[s_][j_], fa1[s_][i_], acc[i_][j_], 0, 0, 0); \
        }
    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < D; ++st) AZ_SLOAD(st);
    // steady state: straight-line body (every slot is used, then refilled D steps ahead), so the waits stay counted
    for (int kt = D; kt < nk; kt += D) {
#pragma unroll
        for (int st = 0; st < D; ++st) { AZ_SMMA(st); AZ_SLOAD(st); }
    }
#pragma unroll
    for (int st = 0; st < D; ++st) AZ_SMMA(st);          // the last D steps: nothing left to fetch
#undef AZ_SLOAD
#undef AZ_SMMA
    // epilogue: a lane holds 4 consecutive output channels of one row
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int mo = m0 + i * 16 + frow;
        if (mo >= M) continue;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int n = n0 + j * 16 + fq * 4;
            const float4 bv = *(const float4*)(d.bias + n);
            float r0 = acc[i][j][0] + bv.x, r1 = acc[i][j][1] + bv.y, r2 = acc[i][j][2] + bv.z, r3 = acc[i][j][3] + bv.w;
            if (d.relu) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); r2 = fmaxf(r2, 0.f); r3 = fmaxf(r3, 0.f); }
            uint2 o;
            o.x = pack_bf16x2(r0, r1);
            o.y = pack_bf16x2(r2, r3);
            *(uint2*)(d.out + (size_t)mo * d.N + n) = o;
        }
    }
}
constexpr int SK_SMALL_WAVES = 256;         // 1 x 1 tiles while they make at most this many waves, 2 x 2 beyond
static inline int skinny_waves_max(int M, int N) {     // a grid that covers either tiling of any batch of up to M output rows
    const int w1 = (std::min(M, SK_SMALL_WAVES * 16 * 16 / N) + 15) / 16 * (N / 16), w2 = (M + 31) / 32 * (N / 32);
    return std::max(std::max(w1, w2), 1);
}
AZ_HD int skinny_waves(int M, int N) {        // waves the launch runs on a batch of M output rows
    const int w1 = (M + 15) / 16 * (N / 16);
    return w1 <= SK_SMALL_WAVES ? w1 : (M + 31) / 32 * (N / 32);
}
template <int LAYER, int D1>      // D1 (ring depth of the 1 x 1 tiles) and D1 / 2 divide K / 64: 8 at C = 512 (72 / 48 / 16 steps), else 6 or 2
__global__ __launch_bounds__(64) void k_gemm_skinny(const GemmDesc d) {
    const int M = (int)(*d.n_dev) * d.rows_per_sample;
    if (M > d.m_max || M <= 0) return;
    const int lane = threadIdx.x;
    if ((M + 15) / 16 * (d.N / 16) <= SK_SMALL_WAVES) {
        const int NT = d.N / 16;
        const int ntile = blockIdx.x % NT, mtile = blockIdx.x / NT;
        if (mtile * 16 >= M) return;
        gemm_skinny_tile<1, 1, D1>(d, M, mtile, ntile, lane);
    } else {
        const int NT = d.N / 32;
        const int ntile = blockIdx.x % NT, mtile = blockIdx.x / NT;
        if (mtile * 32 >= M) return;
        gemm_skinny_tile<2, 2, (D1 >= 4 ? D1 / 2 : D1)>(d, M, mtile, ntile, lane);
    }
}

#ifdef AZ_DIAG
#include "az_net_diag.inc"
#endif

// ---- heads: pi = softmax(x W_pi + b), v = tanh(x w_v + b) (connect_four_net.py:93-95) ---------------------
// one wave per sample; lane holds 8 of the 512 inputs.
__global__ __launch_bounds__(256) void k_heads(const EvalBatch eb, const uint16_t* __restrict__ x /*[n][512] bf16*/,
                                               const float* __restrict__ w /*[8][512]*/, const float* __restrict__ bias /*[8]*/,
                                               uint32_t* __restrict__ n_log /*profile mode: this forward's row count, else nullptr*/) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (n_log && blockIdx.x == 0 && threadIdx.x == 0) *n_log = *eb.n;
    if ((uint32_t)wave >= *eb.n) return;
    const uint4 xv = *(const uint4*)(x + (size_t)wave * 512 + lane * 8);
    float xf[8];
    const uint32_t xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        xf[2 * i] = __uint_as_float(xs[i] << 16);
        xf[2 * i + 1] = __uint_as_float(xs[i] & 0xFFFF0000u);
    }
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float4* wp = (const float4*)(w + (size_t)k * 512 + lane * 8);
        const float4 w0 = wp[0], w1 = wp[1];
        float s = xf[0] * w0.x + xf[1] * w0.y + xf[2] * w0.z + xf[3] * w0.w + xf[4] * w1.x + xf[5] * w1.y + xf[6] * w1.z +
                  xf[7] * w1.w;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        o[k] = s + bias[k];
    }
    if (lane == 0) {
        float mx = o[0];
#pragma unroll
        for (int a = 1; a < ACTIONS; ++a) mx = fmaxf(mx, o[a]);
        float e[ACTIONS], sum = 0.f;
#pragma unroll
        for (int a = 0; a < ACTIONS; ++a) { e[a] = expf(o[a] - mx); sum += e[a]; }
#pragma unroll
        for (int a = 0; a < ACTIONS; ++a) eb.pi[(size_t)wave * 8 + a] = e[a] / sum;
        const float vv = tanhf(o[7]);
        eb.pi[(size_t)wave * 8 + 7] = vv;        // (pi, v) as one 32-byte row: what the backup lanes read
        eb.v[wave] = vv;
    }
}

// ---- host side ------------------------------------------------------------------------------------------
// Parameter vector (f32), "weights file" order -- every conv/FC has a bias (tf.layers defaults) and a BatchNorm
// (gamma, beta, moving_mean, moving_var; eps = 1e-3):
//   conv1 W[3][3][2][C] b[C] bn[4][C] | conv2..4 W[3][3][C][C] b[C] bn[4][C] |
//   fc1 W[6C][1024] b[1024] bn[4][1024] (input index = (y*3+x)*C + c of conv4's [2][3][C] output) |
//   fc2 W[1024][512] b[512] bn[4][512] | pi W[512][7] b[7] | v W[512][1] b[1]
struct ConvNet {                      // the WEIGHTS of one model id (21 MB bf16 + the 20 MB conv1 table at C = 512)
    int C = 512;
    std::vector<float> params;        // raw f32 parameters as set
    std::vector<void*> dev;           // every device allocation
    float *w1 = nullptr, *b1 = nullptr;              // conv1 folded f32 [18][C], [C]
    uint16_t* t1 = nullptr;                            // conv1 table [19683][C] bf16 (k_conv1_table)
    uint16_t* u2 = nullptr;                            // conv2 table [19683][9][C] f16
    uint16_t* w2r = nullptr;                           // conv2's folded weights rearranged [tap*C + co][ci] bf16 (the table GEMM's W)
    uint32_t* npat = nullptr;                          // device constant 19683 (the table GEMM's row count)
    uint16_t* wg[5] = {nullptr};                       // conv2,3,4, fc1, fc2 folded bf16 [N][K]
    uint16_t* wr[5] = {nullptr};                       // conv2,3,4, fc1, fc2 as the ring's stage images (GemmDesc::Wr)
    uint16_t* wp3 = nullptr;                           // conv3's weights as k_conv3_pp's LDS stage images [N / 256][C / 64 * 18][256][32] (C % 256 == 0)
    float* bg[5] = {nullptr};                          // folded bias f32 [N]
    float *wh = nullptr, *bh = nullptr;              // heads f32 [8][512], [8]
    template <class T> T* dalloc(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
        dev.push_back(p);
        return (T*)p;
    }
};

// ---- Conv3Tables: the bank-conflict-free ("PLANES") layout of conv3's LDS image ---------------------------------------------------------
// An A fragment of the image-resident conv3 is 16 CONSECUTIVE OUTPUT positions of a tap: output (y, x) of a 4 x 5 board reads image row
// 7 (y + ky) + (x + kx) of its 6 x 7 board, so the 16 rows of a fragment skip two image rows after every five.  With the image rows in
// order at 128 bytes each and the usual chunk ^ (row & 7) swizzle, a ds_read_b128 lane group (16 lanes: MI355X_MICROARCH.md, LDS) finds
// rows r and r + 8 on the same 16-byte bank slot: 1.94 LDS cycles per group instead of 1 on average over the taps and tiles, the 38 % of
// LDS cycles round 2's SQ_LDS_BANK_CONFLICT counter showed; no swizzle of whole 128-byte rows avoids it for all nine taps (a lane group
// mixes two k chunks, and the row set moves by one slot from tap to tap).  The layout here gives every image row the LABEL
// o' = 20 board + 5 y' + x' (mod 16) -- the output index a position would have, so the 16 rows of any fragment of any tap carry 16
// consecutive labels -- and places the row where its slot IS its label:
//   * two PLANES by the parity of the 16-byte k chunk (the two chunks a lane group mixes differ in exactly that bit; a plane is 32 KiB,
//     a multiple of the 256-byte bank row, so both land on the same slots);
//   * in a plane a row is one 64-byte cell (4 chunks); cell index rho = 4 m + (label >> 2), chunk position = pair index ^ (label & 3):
//     slot = (rho & 3) * 4 + position = the label, XORed with a constant of the instruction.  16 distinct labels = 16 distinct slots.
//   * m numbers the rows of a label class in image order: 126 rows per class, 504 of the 512 cells used -- 64 KiB, which with the
//     16 KiB weight buffer is exactly half of a CU's LDS (two workgroups per CU, as before).
// No closed form is needed on the device: the DMA side reads inv[rho] = (image row, label & 3) once per kernel (8 rows per lane), the
// fragment side streams rd[tap][wave row][fragment row][row tile] = the cell's byte offset, one 16-byte entry per lane and K-step,
// fetched a step ahead.  The global side of the DMA reads 64 contiguous bytes per image row and instruction instead of 128.
static std::vector<uint16_t> conv3_tables() {
    constexpr int NB = C3_NB, IH = 6, IW = 7, OW = 5, OUT_PER = 20, IN_PER = 42;
    std::vector<uint16_t> t((size_t)C3_TAB_INV + C3_TAB_RD, 0);
    int next_m[4] = {0, 0, 0, 0};
    std::vector<int> cell(NB * IN_PER, 0);            // rho * 64 + (label & 3) * 16 per image row
    for (int bl = 0; bl < NB; ++bl)
        for (int y = 0; y < IH; ++y)
            for (int x = 0; x < IW; ++x) {
                const int label = (OUT_PER * bl + OW * y + x) & 15, cls = label >> 2, h2 = label & 3;
                const int rho = 4 * next_m[cls]++ + cls;
                if (rho >= C3_TAB_INV) return {};
                const int r = bl * IN_PER + y * IW + x;
                t[rho] = (uint16_t)(r | (h2 << 10));
                cell[r] = rho * 64 + h2 * 16;
            }
    for (int tap = 0; tap < 9; ++tap)
        for (int wr = 0; wr < 2; ++wr)
            for (int frow = 0; frow < 16; ++frow)
                for (int mt = 0; mt < 8; ++mt) {
                    int ml = wr * 128 + mt * 16 + frow;
                    if (ml >= NB * OUT_PER) ml = frow;        // rows past the tile (never stored): 16 distinct valid cells, conflict-free too
                    const int bl = ml / OUT_PER, p = ml % OUT_PER, y = p / OW + tap / 3, x = p % OW + tap % 3;
                    t[(size_t)C3_TAB_INV + ((tap * 2 + wr) * 16 + frow) * 8 + mt] = (uint16_t)cell[bl * IN_PER + y * IW + x];
                }
    return t;
}

// Activation workspace of ONE stream (1.2 GB at 8192 rows, C = 512), shared by every model that runs on that stream:
// model ids come and go with the Coach loop (src/coach.rs:296-390), the workspace does not grow with them.
struct NetWorkspace {
    int C = 512, max_batch = 0;
    std::vector<void*> dev;
    uint16_t *act1 = nullptr, *act2 = nullptr, *act3 = nullptr, *act4 = nullptr, *fc1o = nullptr, *fc2o = nullptr;
    // profiling: event quads per forward + pinned copies of the batch size
    struct Rec { hipEvent_t e0, e1, e2, e2b, e2c, e3; uint32_t* n; int table2; };   // e1..e2 conv2, e2..e2b conv3, e2b..e2c conv4, e2c..e3 fc1 + fc2 + heads
    std::vector<Rec> open;
    std::vector<hipEvent_t> ev_pool;
    uint32_t* pinned_n = nullptr;          // host copy of d_nlog (one copy per resolve, not one per forward)
    uint32_t* d_nlog = nullptr;            // [pinned_cap] row count of every timed forward, written by its k_heads
    int pinned_cap = 0, pinned_next = 0;
    unsigned long long* dbg = nullptr;     // [2048] clock stamps of the diagnostic variant
    uint16_t* c3tab = nullptr;             // Conv3Tables (the PLANES layout of conv3's LDS image)
    unsigned long long* acct = nullptr;    // [2] k_conv3_auto's rows and working launches
    template <class T> T* dalloc(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
        dev.push_back(p);
        return (T*)p;
    }
};

int64_t convnet_param_count(int channels) { return Layout(channels).total; }

ConvNet* convnet_create(int channels, const char** err) {
    if (channels % 128 != 0 || channels < 128) { if (err) *err = "net_channels must be a multiple of 128"; return nullptr; }
    ConvNet* n = new ConvNet();
    n->C = channels;
    const int C = channels;
    bool ok = true;
    ok &= (n->w1 = n->dalloc<float>(18 * (size_t)C)) != nullptr;
    ok &= (n->b1 = n->dalloc<float>(C)) != nullptr;
    ok &= (n->t1 = n->dalloc<uint16_t>((size_t)CONV1_PATTERNS * C)) != nullptr;
    ok &= (n->u2 = n->dalloc<uint16_t>((size_t)(CONV1_PATTERNS + 1) * 9 * C)) != nullptr;     // + one all-zero row (k_conv2_table_x)
    if (n->u2) ok &= hipMemset(n->u2 + (size_t)CONV1_PATTERNS * 9 * C, 0, (size_t)9 * C * sizeof(uint16_t)) == hipSuccess;
    ok &= (n->w2r = n->dalloc<uint16_t>((size_t)9 * C * C)) != nullptr;
    ok &= (n->npat = n->dalloc<uint32_t>(1)) != nullptr;
    if (ok) { const uint32_t np = CONV1_PATTERNS; ok = hipMemcpy(n->npat, &np, sizeof np, hipMemcpyHostToDevice) == hipSuccess; }
    const size_t wk[5] = {9 * (size_t)C, 9 * (size_t)C, 9 * (size_t)C, 6 * (size_t)C, 1024};
    const size_t wn[5] = {(size_t)C, (size_t)C, (size_t)C, 1024, 512};
    for (int l = 0; l < 5; ++l) {
        ok &= (n->wg[l] = n->dalloc<uint16_t>(wk[l] * wn[l])) != nullptr;
        ok &= (n->bg[l] = n->dalloc<float>(wn[l])) != nullptr;
        if (l >= 1) ok &= (n->wr[l] = n->dalloc<uint16_t>(wk[l] * wn[l])) != nullptr;
    }
    ok &= (n->wh = n->dalloc<float>(8 * 512)) != nullptr;
    ok &= (n->bh = n->dalloc<float>(8)) != nullptr;
#ifdef AZ_DIAG
    if (C % PP_NCOL == 0) ok &= (n->wp3 = n->dalloc<uint16_t>(9 * (size_t)C * C)) != nullptr;
#endif
    if (!ok) {
        if (err) *err = "convnet_create: device allocation failed";
        convnet_destroy(n);
        return nullptr;
    }
    return n;
}

void convnet_destroy(ConvNet* n) {
    if (!n) return;
    for (void* p : n->dev) (void)hipFree(p);
    delete n;
}

NetWorkspace* netws_create(int channels, int max_batch, const char** err) {
    NetWorkspace* n = new NetWorkspace();
    n->C = channels;
    n->max_batch = max_batch;
    const int C = channels;
    const size_t B = (size_t)max_batch;
    bool ok = true;
    ok &= (n->act2 = n->dalloc<uint16_t>((B + 16) * 42 * C)) != nullptr;     // + 16 boards: k_conv_valid_pipe reads its last tile unclamped
    ok &= (n->act3 = n->dalloc<uint16_t>(B * 20 * C)) != nullptr;
    ok &= (n->act4 = n->dalloc<uint16_t>(B * 6 * C)) != nullptr;
    ok &= (n->fc1o = n->dalloc<uint16_t>(B * 1024)) != nullptr;
    ok &= (n->fc2o = n->dalloc<uint16_t>(B * 512)) != nullptr;
    ok &= (n->dbg = n->dalloc<unsigned long long>(2048)) != nullptr;
    if (ok) ok = hipMemset(n->dbg, 0, 2048 * 8) == hipSuccess;
    ok &= (n->acct = n->dalloc<unsigned long long>(2)) != nullptr;
    if (ok) ok = hipMemset(n->acct, 0, 16) == hipSuccess;
    ok &= (n->c3tab = n->dalloc<uint16_t>(C3_TAB_INV + C3_TAB_RD)) != nullptr;
    if (ok) {
        const std::vector<uint16_t> tab = conv3_tables();
        ok = !tab.empty() && hipMemcpy(n->c3tab, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok) ok = hipHostMalloc((void**)&n->pinned_n, 4096 * sizeof(uint32_t)) == hipSuccess;
    ok &= (n->d_nlog = n->dalloc<uint32_t>(4096)) != nullptr;
    n->pinned_cap = 4096;
    if (!ok) {
        if (err) *err = "netws_create: device allocation failed";
        netws_destroy(n);
        return nullptr;
    }
    return n;
}

// act1 (conv1's haloed output image, 604 MB at 8192 rows) is only needed by the kernel sets that run conv1 as a kernel
static bool netws_need_act1(NetWorkspace* n) {
    if (n->act1) return true;
    const size_t bytes = (size_t)n->max_batch * 72 * n->C * sizeof(uint16_t);
    n->act1 = n->dalloc<uint16_t>(bytes / sizeof(uint16_t));
    return n->act1 && hipMemset(n->act1, 0, bytes) == hipSuccess;   // the zero halo
}

bool convnet_prepare(NetWorkspace* ws, const NetOptions& o) {
    const bool shipped_set = o.gemm_variant == 5;
    const bool table2 = o.conv2_table && shipped_set;
    const bool table = !table2 && o.conv1_table && shipped_set && ws->C % HBN_ == 0;
    return (table || table2) ? true : netws_need_act1(ws);
}

void netws_destroy(NetWorkspace* n) {
    if (!n) return;
    for (void* p : n->dev) (void)hipFree(p);
    for (auto& r : n->open) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); (void)hipEventDestroy(r.e2); (void)hipEventDestroy(r.e2b); (void)hipEventDestroy(r.e2c); (void)hipEventDestroy(r.e3); }
    for (auto e : n->ev_pool) (void)hipEventDestroy(e);
    if (n->pinned_n) (void)hipHostFree(n->pinned_n);
    delete n;
}

// fold BatchNorm (inference) into the producing layer: w' = w * g/sqrt(var+eps), b' = (b-mean)*g/sqrt(var+eps) + beta
static void fold_scale(const float* bn, int n, int idx, float* scale, float* shift_mul_mean) {
    const float gamma = bn[idx], beta = bn[n + idx], mean = bn[2 * n + idx], var = bn[3 * n + idx];
    const float s = gamma / std::sqrt(var + 1e-3f);
    *scale = s;
    *shift_mul_mean = beta - mean * s;
}

bool convnet_set_params(ConvNet* net, const float* p, int64_t count) {
    const Layout L(net->C);
    if (count != L.total) return false;
    net->params.assign(p, p + count);
    const int C = net->C;
    bool ok = true;
    {   // conv1: f32 [18][C], k = (ky*3+kx)*2 + ci
        std::vector<float> w(18 * (size_t)C), b(C);
        for (int co = 0; co < C; ++co) {
            float s, t;
            fold_scale(p + L.conv_bn[0], C, co, &s, &t);
            for (int k = 0; k < 18; ++k) w[(size_t)k * C + co] = p[L.conv_w[0] + (int64_t)k * C + co] * s;
            b[co] = p[L.conv_b[0] + co] * s + t;
        }
        ok &= hipMemcpy(net->w1, w.data(), w.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        ok &= hipMemcpy(net->b1, b.data(), b.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        hipLaunchKernelGGL(k_conv1_table, dim3(CONV1_PATTERNS), dim3(64), 0, nullptr, net->w1, net->b1, net->t1, C);
        ok &= hipDeviceSynchronize() == hipSuccess;
    }
    // conv2..4 and fc1, fc2: bf16 [N][K] (K index = tap*cin + ci = the raw weight's leading index)
    const int64_t w_off[5] = {L.conv_w[1], L.conv_w[2], L.conv_w[3], L.fc_w[0], L.fc_w[1]};
    const int64_t b_off[5] = {L.conv_b[1], L.conv_b[2], L.conv_b[3], L.fc_b[0], L.fc_b[1]};
    const int64_t bn_off[5] = {L.conv_bn[1], L.conv_bn[2], L.conv_bn[3], L.fc_bn[0], L.fc_bn[1]};
    const int K[5] = {9 * C, 9 * C, 9 * C, 6 * C, 1024}, N[5] = {C, C, C, 1024, 512};
    for (int l = 0; l < 5; ++l) {
        std::vector<uint16_t> w((size_t)K[l] * N[l]);
        std::vector<float> b(N[l]);
        for (int nn = 0; nn < N[l]; ++nn) {
            float s, t;
            fold_scale(p + bn_off[l], N[l], nn, &s, &t);
            b[nn] = p[b_off[l] + nn] * s + t;
            for (int k = 0; k < K[l]; ++k) w[(size_t)nn * K[l] + k] = f32_to_bf16_host(p[w_off[l] + (int64_t)k * N[l] + nn] * s);
        }
        ok &= hipMemcpy(net->wg[l], w.data(), w.size() * 2, hipMemcpyHostToDevice) == hipSuccess;
        ok &= hipMemcpy(net->bg[l], b.data(), b.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        if (net->wr[l]) {   // the same bf16 values as the ring's stage images: K-steps in the walker's order (channel block outer, tap inner)
            std::vector<uint16_t> wr(w.size());
            const int cin = l <= 2 ? C : K[l], ntaps = K[l] / cin, nk = K[l] / 64;
            for (int nt = 0; nt < N[l] / 128; ++nt)
                for (int kt = 0; kt < nk; ++kt) {
                    const int cb = kt / ntaps, tap = kt - cb * ntaps, kk = tap * cin + cb * 64;
                    uint16_t* img = &wr[((size_t)nt * nk + kt) * 8192];
                    for (int r = 0; r < 128; ++r)
                        for (int sl = 0; sl < 8; ++sl)
                            std::memcpy(img + r * 64 + sl * 8, &w[(size_t)(nt * 128 + r) * K[l] + kk + ((sl ^ (r & 7)) << 3)], 16);
                }
            ok &= hipMemcpy(net->wr[l], wr.data(), wr.size() * 2, hipMemcpyHostToDevice) == hipSuccess;
        }
#ifdef AZ_DIAG
        if (l == 1 && net->wp3) {   // conv3: the same bf16 values as k_conv3_pp's stage images (stage = channel block, tap, k half)
            std::vector<uint16_t> wp((size_t)9 * C * C);
            const int nst = C / 64 * 18;
            for (int nt = 0; nt < C / PP_NCOL; ++nt)
                for (int st = 0; st < nst; ++st) {
                    const int cb = st / 18, tap = (st % 18) / 2, half = st & 1;
                    uint16_t* img = &wp[((size_t)nt * nst + st) * (PP_WSTAGE / 2)];
                    for (int nl = 0; nl < PP_NCOL; ++nl)
                        for (int q = 0; q < 4; ++q)
                            std::memcpy(img + nl * 32 + ((q ^ pp_wperm(nl)) << 3),
                                        &w[(size_t)(nt * PP_NCOL + nl) * K[1] + (size_t)tap * C + cb * 64 + half * 32 + q * 8], 16);
                }
            ok &= hipMemcpy(net->wp3, wp.data(), wp.size() * 2, hipMemcpyHostToDevice) == hipSuccess;
        }
#endif
        if (l == 0) {   // conv2: the same bf16 values as [tap*C + co][ci] for the table GEMM, then U = T x W2r^T
            std::vector<uint16_t> wr((size_t)9 * C * C);
            for (int co = 0; co < C; ++co)
                for (int t = 0; t < 9; ++t)
                    std::memcpy(&wr[((size_t)t * C + co) * C], &w[(size_t)co * K[0] + (size_t)t * C], (size_t)C * 2);
            ok &= hipMemcpy(net->w2r, wr.data(), wr.size() * 2, hipMemcpyHostToDevice) == hipSuccess;
            GemmDesc d{};
            d.A = net->t1; d.W = net->w2r; d.bias = net->bg[0]; d.out = net->u2; d.n_dev = net->npat;
            d.rows_per_sample = 1; d.out_w = 1; d.in_h = 1; d.in_w = 1; d.in_c = C; d.tap_w = 1; d.cin = C; d.K = C; d.N = 9 * C;
            d.relu = 0; d.out_f16 = 1;
            const int mt8 = ((CONV1_PATTERNS + GBM - 1) / GBM + 7) / 8 * 8;
            hipLaunchKernelGGL(k_gemm_mfma<6>, dim3(mt8 * (d.N / GBN)), dim3(256), 0, nullptr, d);
            ok &= hipDeviceSynchronize() == hipSuccess;
        }
    }
    {   // heads: f32 [8][512]
        std::vector<float> w(8 * 512), b(8);
        for (int k = 0; k < 512; ++k) {
            for (int a = 0; a < 7; ++a) w[(size_t)a * 512 + k] = p[L.pi_w + (int64_t)k * 7 + a];
            w[(size_t)7 * 512 + k] = p[L.v_w + k];
        }
        for (int a = 0; a < 7; ++a) b[a] = p[L.pi_b + a];
        b[7] = p[L.v_b];
        ok &= hipMemcpy(net->wh, w.data(), w.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        ok &= hipMemcpy(net->bh, b.data(), b.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    return ok;
}

bool convnet_get_params(const ConvNet* n, float* host_params, int64_t count) {
    if (!n || (int64_t)n->params.size() != count) return false;
    std::memcpy(host_params, n->params.data(), (size_t)count * sizeof(float));
    return true;
}

// Glorot-uniform kernels, zero biases, BN gamma=1 beta=0 mean=0 var=1 (the tf.layers defaults the reference
// would start from, connect_four_net.py:36-100).  Uniforms come from the build's counter RNG.
void convnet_init_random(ConvNet* net, uint64_t seed) {
    const Layout L(net->C);
    const int C = net->C;
    std::vector<float> p((size_t)L.total, 0.0f);
    uint64_t ctr = 0;
    auto fill = [&](int64_t off, int64_t n, int fan_in, int fan_out) {
        const float lim = std::sqrt(6.0f / (float)(fan_in + fan_out));
        for (int64_t i = 0; i < n; ++i) {
            uint64_t r = rng_draw(seed, ctr++, 0, RNG_WEIGHTS);
            float u = (float)(uint32_t)(r >> 40) * (1.0f / 16777216.0f);   // [0,1)
            p[(size_t)(off + i)] = (2.0f * u - 1.0f) * lim;
        }
    };
    auto bn_default = [&](int64_t off, int n) {
        for (int i = 0; i < n; ++i) { p[(size_t)(off + i)] = 1.0f; p[(size_t)(off + 3 * n + i)] = 1.0f; }
    };
    for (int l = 0; l < 4; ++l) {
        int cin = l == 0 ? 2 : C;
        fill(L.conv_w[l], 9ll * cin * C, 9 * cin, 9 * C);
        bn_default(L.conv_bn[l], C);
    }
    fill(L.fc_w[0], 6ll * C * 1024, 6 * C, 1024); bn_default(L.fc_bn[0], 1024);
    fill(L.fc_w[1], 1024ll * 512, 1024, 512); bn_default(L.fc_bn[1], 512);
    fill(L.pi_w, 512 * 7, 512, 7);
    fill(L.v_w, 512, 512, 1);
    convnet_set_params(net, p.data(), L.total);
}

// ---- kernel choice per layer ----------------------------------------------------------------------------------------------------
// rows_hint = upper bound on the batch (the grids must cover it: tiles past the device-side count exit at once);
// rows_typ = what the batch is expected to hold (picks the kernel / tile family; any value is correct, a good one is fast).
// Rejected and removed after measurement (numbers in profiles/README.md, code in git history): XCD column remap, third weight buffer,
// late / spread DMA issue, mid barrier, 32x32x16 MFMA shape, non-temporal cache policy, wave stagger, persistent tiles, tail split,
// cross-step fragment prefetch, weights straight into registers, a double-buffered conv3 image, one wave per SIMD.

// Small batches (the arena, the drain of a self-play call, single-tree calls): the image-resident conv3 kernel is a chain of 72 K-steps of
// ~0.8 us for a workgroup alone on its CU (70 us whatever the rows); the ring with 4 stages in flight walks the same K in ~32 us up to
// 128 rows and 49 us at 384 (tools/rows_sweep.py).  Taken when the expected rows fit one workgroup per CU.
static bool conv3_is_small(const GemmDesc& d, int rows_hint, int rows_typ) {
    return (rows_typ > 0 ? (long long)rows_typ * 115 / 100 : (long long)rows_hint) * d.rows_per_sample <= 8192;      // no estimate: the bound itself
}
// conv2 as a GEMM ("conv2_table" = 0), image-resident
template <bool TABLE>
static void launch_conv2_gemm(const GemmDesc& d, int rows_hint, hipStream_t s) {
    const int tiles = (rows_hint + IMG_NB - 1) / IMG_NB;
    const int t8 = (tiles + 7) / 8 * 8;
    hipLaunchKernelGGL((k_conv_same_pipe<1, TABLE>), dim3(t8 * (d.N / HBN2_)), dim3(256), 0, s, d);
}
static void launch_conv3_image(const GemmDesc& d, int rows_hint, hipStream_t s, bool tail, bool planes, int pp) {
#ifdef AZ_DIAG
    if (pp && d.Wp && d.N % PP_NCOL == 0) {
        const int t8 = ((rows_hint + PP_NB - 1) / PP_NB + 7) / 8 * 8;
        const dim3 grid(t8 * (d.N / PP_NCOL)), block(512);
        switch (pp) {           // 16 + mask: the timing ablations (WRONG results)
            case 17: hipLaunchKernelGGL((k_conv3_pp<2, 1>), grid, block, 0, s, d); return;
            case 18: hipLaunchKernelGGL((k_conv3_pp<2, 2>), grid, block, 0, s, d); return;
            case 19: hipLaunchKernelGGL((k_conv3_pp<2, 3>), grid, block, 0, s, d); return;
            case 20: hipLaunchKernelGGL((k_conv3_pp<2, 4>), grid, block, 0, s, d); return;
            case 23: hipLaunchKernelGGL((k_conv3_pp<2, 7>), grid, block, 0, s, d); return;
            case 24: hipLaunchKernelGGL((k_conv3_pp<2, 8>), grid, block, 0, s, d); return;
            case 27: hipLaunchKernelGGL((k_conv3_pp<2, 11>), grid, block, 0, s, d); return;
            case 28: hipLaunchKernelGGL((k_conv3_pp<2, 16>), grid, block, 0, s, d); return;
            case 34: hipLaunchKernelGGL((k_conv3_pp<2, 0, 1>), grid, block, 0, s, d); return;
            case 36: hipLaunchKernelGGL((k_conv3_pp<2, 128>), grid, block, 0, s, d); return;
            case 38: hipLaunchKernelGGL((k_conv3_pp<2, 0, 2, 0>), grid, block, 0, s, d); return;
            case 39: hipLaunchKernelGGL((k_conv3_pp<2, 0, 2, 1>), grid, block, 0, s, d); return;
            case 40: hipLaunchKernelGGL((k_conv3_pp<2, 0, 2, 3>), grid, block, 0, s, d); return;
            case 37: hipLaunchKernelGGL((k_conv3_pp<2, 128 + 16>), grid, block, 0, s, d); return;
            case 35: hipLaunchKernelGGL((k_conv3_pp<2, 16, 1>), grid, block, 0, s, d); return;
            case 30: hipLaunchKernelGGL((k_conv3_pp<2, 32>), grid, block, 0, s, d); return;
            case 31: hipLaunchKernelGGL((k_conv3_pp<2, 64>), grid, block, 0, s, d); return;
            case 32: hipLaunchKernelGGL((k_conv3_pp<2, 48>), grid, block, 0, s, d); return;
            case 33: hipLaunchKernelGGL((k_conv3_pp<2, 80>), grid, block, 0, s, d); return;
            case 29: hipLaunchKernelGGL((k_conv3_pp<2, 24>), grid, block, 0, s, d); return;
            default: break;
        }
        hipLaunchKernelGGL((k_conv3_pp<2>), grid, block, 0, s, d);
        return;
    }
#endif
    const int tiles = (rows_hint + C3_NB - 1) / C3_NB;
    const int t8 = (tiles + 7) / 8 * 8;
    const int full_grid = t8 * (d.N / 128);
    if (planes) hipLaunchKernelGGL((k_conv3_auto<2, true>), dim3(full_grid + (tail ? C3_TAIL : 0)), dim3(256), 0, s, d, full_grid, tail ? C3_TAIL : 0);
    else if (tail) hipLaunchKernelGGL((k_conv3_auto<2, false>), dim3(full_grid + C3_TAIL), dim3(256), 0, s, d, full_grid, C3_TAIL);
    else hipLaunchKernelGGL((k_conv_valid_pipe<2, C3_NB, 6, 7, false, 0, true>), dim3(full_grid), dim3(256), 0, s, d);
}

// The LDS-DMA ring with the tile rows picked on the device.  The host picks the FAMILY from its estimate (NS = 4: one workgroup per CU,
// for grids of at most 256 tiles; the estimate + 15 %: a batch over the limit would pay a whole second round), the kernel picks the
// tile rows from the exact count (measured: tools/ring_tiles.py, profiles/README.md).
template <int LAYER>
static void launch_ring_auto(const GemmDesc& d, int rows_hint, int rows_typ, hipStream_t s, bool force_one_per_cu = false) {
    const int m_est = (int)((rows_typ > 0 ? (long long)rows_typ * 115 / 100 : (long long)rows_hint) * d.rows_per_sample), ncol = d.N / GBN;
    const bool conv = d.tap_w > 1;
    const int tile_max = conv ? 96 : 128;                                    // the largest tile of the one-workgroup-per-CU family
    const int m_one = 256 / ncol * tile_max;                                 // the most rows that family covers in ONE round
    const int m_bound = rows_hint * d.rows_per_sample;
    auto launch = [&](bool one_per_cu, const GemmDesc& dd, int m_cover) {
        const int bmin = one_per_cu ? 64 : 96;                               // the grid covers the smallest tile of the family
        const int mtb = ((m_cover + bmin - 1) / bmin + 7) / 8 * 8;
        if (one_per_cu) hipLaunchKernelGGL((k_gemm_ring_auto<LAYER, 4>), dim3(mtb * ncol), dim3(256), 0, s, dd);
        else hipLaunchKernelGGL((k_gemm_ring_auto<LAYER, 2>), dim3(mtb * ncol), dim3(256), 0, s, dd);
    };
    // The estimate (the previous move's largest batch) cannot tell the families apart near the line: a batch just under it on the two-per-CU
    // family is 43 instead of ~30 us for fc1, one just over it on the one-per-CU family a whole second round.  Near the line BOTH are
    // launched and each checks the batch's exact row count on the device (m_hi / m_min): one runs, the other exits at once.
    if (!force_one_per_cu && !conv && m_bound > m_one && (long long)m_est * 100 >= (long long)m_one * 70 && (long long)m_est * 100 <= (long long)m_one * 140) {
        GemmDesc lo = d, hi = d;
        lo.m_hi = m_one;
        hi.m_min = std::max(d.m_min, m_one);
        launch(true, lo, std::min(m_bound, m_one));
        launch(false, hi, m_bound);
        return;
    }
    const bool one_per_cu = force_one_per_cu || m_bound <= m_one || (m_est + tile_max - 1) / tile_max * ncol <= 256;     // the bound itself fits one round
    launch(one_per_cu, d, m_bound);
}

#ifdef AZ_DIAG
// The diagnostic library's launcher: every kernel generation and forced tile behind the NetOptions switches.  Returns false when the
// options ask for nothing special, i.e. the shipped choice below applies.
template <int LAYER>
static bool launch_gemm_diag(const GemmDesc& d, int rows_hint, int rows_typ, hipStream_t s, const NetOptions& o) {
    const int v = o.gemm_variant;
    if ((v == 3 || v == 5) && LAYER == 1 && d.N % HBN_ == 0 && d.cin % 64 == 0) {
        const int tiles = (rows_hint + IMG_NB - 1) / IMG_NB;
        const int t8 = (tiles + 7) / 8 * 8;
        if (v == 5 && o.conv2_pipe) return false;
        if (v == 5 && d.states) hipLaunchKernelGGL((k_conv_img2<LAYER, true>), dim3(t8 * (d.N / HBN2_)), dim3(256), 0, s, d);
        else if (v == 5) hipLaunchKernelGGL((k_conv_img2<LAYER>), dim3(t8 * (d.N / HBN2_)), dim3(256), 0, s, d);
        else hipLaunchKernelGGL((k_conv_img<LAYER>), dim3(t8 * (d.N / HBN_)), dim3(512), 0, s, d);
        return true;
    }
    const bool big = v >= 1 && (LAYER == 1 || LAYER == 2 || (LAYER == 3 && !o.ring_tile[3] && (o.conv4_big == 1 || (o.conv4_big == 2 && rows_typ >= 4096)))) &&
                     d.N % HBN_ == 0;
    if constexpr (LAYER == 2) if (v == 5 && o.conv3_ring) {      // conv3 forced onto the ring (im2col from act2): 1 / 2 = 128-row tiles with 2 / 4 stages, 3 = device-picked tile
        const int mt = (rows_hint * d.rows_per_sample + GBM - 1) / GBM;
        const int mt8 = (mt + 7) / 8 * 8;
        if (o.conv3_ring == 3) launch_ring_auto<LAYER>(d, rows_hint, rows_typ, s, true);
        else if (o.conv3_ring == 2) hipLaunchKernelGGL((k_gemm_ring<LAYER, 4>), dim3(mt8 * (d.N / GBN)), dim3(256), 0, s, d);
        else hipLaunchKernelGGL((k_gemm_ring<LAYER, 2>), dim3(mt8 * (d.N / GBN)), dim3(256), 0, s, d);
        return true;
    }
    if constexpr (LAYER == 2) if (v == 5 && d.rows_per_sample == 20) {   // conv3 image-resident (or, for a small expected batch, the shipped ring choice)
        if ((o.conv3_small && conv3_is_small(d, rows_hint, rows_typ)) || o.conv3_pipe == 1) return false;
        const int tiles = (rows_hint + C3_NB - 1) / C3_NB;
        const int t8 = (tiles + 7) / 8 * 8;
        const dim3 g3(t8 * (d.N / 128)), b3(256);
        switch (o.conv3_pipe) {
            case 2: hipLaunchKernelGGL((k_conv_valid_pipe<LAYER, C3_NB, 6, 7, false, 0, false>), g3, b3, 0, s, d); break;
            case 3: hipLaunchKernelGGL((k_conv_valid_pipe<LAYER, C3_NB, 6, 7, true, 0, true>), g3, b3, 0, s, d); break;
            case 10: hipLaunchKernelGGL((k_conv_valid_pipe<LAYER, C3_NB, 6, 7, false, -1, true>), g3, b3, 0, s, d); break;
            case 9: hipLaunchKernelGGL((k_conv_valid_pipe<LAYER, C3_NB, 6, 7, false, -2, true>), g3, b3, 0, s, d); break;
            case 11: hipLaunchKernelGGL((k_conv_valid_pipe<LAYER, C3_NB, 6, 7, false, 1, true>), g3, b3, 0, s, d); break;
            case 12: hipLaunchKernelGGL((k_conv_valid_pipe<LAYER, C3_NB, 6, 7, false, 2, true>), g3, b3, 0, s, d); break;
            case 13: hipLaunchKernelGGL((k_conv_valid_pipe<LAYER, C3_NB, 6, 7, false, 3, true>), g3, b3, 0, s, d); break;
            case 14: hipLaunchKernelGGL((k_conv_valid_pipe<LAYER, C3_NB, 6, 7, false, 4, true>), g3, b3, 0, s, d); break;
            case 15: hipLaunchKernelGGL((k_conv_valid_pipe<LAYER, C3_NB, 6, 7, false, 5, true>), g3, b3, 0, s, d); break;
            default: hipLaunchKernelGGL((k_conv_valid_img2<LAYER, C3_NB, 6, 7, 2>), g3, b3, 0, s, d); break;
        }
        return true;
    }
    if (v == 5 && !big && o.fc_ring == 1 && !(LAYER >= 3 && o.ring_tile[LAYER])) return false;
    if (big) {
        const int mt = (rows_hint * d.rows_per_sample + HBM_ - 1) / HBM_;
        const int mt8 = (mt + 7) / 8 * 8;
        const dim3 grid(mt8 * (d.N / HBN_)), block(512);
        if (v == 1) hipLaunchKernelGGL((k_gemm256<LAYER, 0>), grid, block, 0, s, d);
        else if (v == 11) hipLaunchKernelGGL((k_gemm256<LAYER, 1, 1>), grid, block, 0, s, d);
        else if (v == 12) hipLaunchKernelGGL((k_gemm256<LAYER, 1, 2>), grid, block, 0, s, d);
        else if (v == 13) hipLaunchKernelGGL((k_gemm256<LAYER, 1, 3>), grid, block, 0, s, d);
        else if (v == 14) hipLaunchKernelGGL((k_gemm256<LAYER, 1, 4>), grid, block, 0, s, d);
        else if (v == 15) hipLaunchKernelGGL((k_gemm256<LAYER, 1, 5>), grid, block, 0, s, d);
        else if (v == 16) hipLaunchKernelGGL((k_gemm256<LAYER, 1, 6>), grid, block, 0, s, d);
        else if (v == 17) hipLaunchKernelGGL((k_gemm256<LAYER, 1, 7>), grid, block, 0, s, d);
        else hipLaunchKernelGGL((k_gemm256<LAYER, 1>), grid, block, 0, s, d);
        return true;
    }
    const int mt = (rows_hint * d.rows_per_sample + GBM - 1) / GBM;
    const int mt8 = (mt + 7) / 8 * 8;
    const int grid = mt8 * (d.N / GBN);
    if (v == 5 && o.fc_ring) {
        auto ring = [&](auto bm_c, auto ns_c) {
            constexpr int BM = decltype(bm_c)::value, NS = decltype(ns_c)::value;
            const int mtb = ((rows_hint * d.rows_per_sample + BM - 1) / BM + 7) / 8 * 8;
            hipLaunchKernelGGL((k_gemm_ring<LAYER, NS, BM>), dim3(mtb * (d.N / GBN)), dim3(256), 0, s, d);
        };
        using std::integral_constant;
        if constexpr (LAYER >= 3) {
            switch (o.ring_tile[LAYER]) {
                case 642: ring(integral_constant<int, 64>{}, integral_constant<int, 2>{}); return true;
                case 644: ring(integral_constant<int, 64>{}, integral_constant<int, 4>{}); return true;
                case 962: ring(integral_constant<int, 96>{}, integral_constant<int, 2>{}); return true;
                case 964: ring(integral_constant<int, 96>{}, integral_constant<int, 4>{}); return true;
                case 1282: ring(integral_constant<int, 128>{}, integral_constant<int, 2>{}); return true;
                case 1284: ring(integral_constant<int, 128>{}, integral_constant<int, 4>{}); return true;
                case 1602: ring(integral_constant<int, 160>{}, integral_constant<int, 2>{}); return true;
                case 1922: ring(integral_constant<int, 192>{}, integral_constant<int, 2>{}); return true;
                default: break;
            }
        }
        if (o.fc_ring == 2) {       // A/B: the tile picked on the HOST from its estimate of the row count (same rule)
            const int m_typ = (rows_typ > 0 ? rows_typ : rows_hint) * d.rows_per_sample, ncol = d.N / GBN;
            const bool conv = d.tap_w > 1;
            const int ns = (m_typ + (conv ? 95 : 127)) / (conv ? 96 : 128) * ncol <= 256 ? 4 : 2;
            switch (ring_pick_bm(m_typ, ncol, ns, conv) * 10 + ns) {
                case 644: ring(integral_constant<int, 64>{}, integral_constant<int, 4>{}); break;
                case 964: ring(integral_constant<int, 96>{}, integral_constant<int, 4>{}); break;
                case 1284: ring(integral_constant<int, 128>{}, integral_constant<int, 4>{}); break;
                case 962: ring(integral_constant<int, 96>{}, integral_constant<int, 2>{}); break;
                case 1602: ring(integral_constant<int, 160>{}, integral_constant<int, 2>{}); break;
                case 1922: ring(integral_constant<int, 192>{}, integral_constant<int, 2>{}); break;
                default: ring(integral_constant<int, 128>{}, integral_constant<int, 2>{}); break;
            }
            return true;
        }
        if (o.fc_ring == 1) return false;
        hipLaunchKernelGGL((k_gemm_ring<LAYER, 2>), dim3(grid), dim3(256), 0, s, d);     // "fc_ring" = 3: the plain 128-row ring
        return true;
    }
    hipLaunchKernelGGL(k_gemm_mfma<LAYER>, dim3(grid), dim3(256), 0, s, d);
    return true;
}
#endif

template <int LAYER>
static void launch_gemm(const GemmDesc& d, int rows_hint, int rows_typ, hipStream_t s, const NetOptions& o) {
#ifdef AZ_DIAG
    if (launch_gemm_diag<LAYER>(d, rows_hint, rows_typ, s, o)) return;
#endif
    // Small batches: the register-fed skinny GEMM (k_gemm_skinny) for batches of at most `lim` boards (per layer: a conv3 row is 20
    // output rows, a conv4 row 6, an fc row 1).  The host only knows an estimate (the largest batch of the previous move), and in the arena
    // or a draining self-play call most steps are far smaller than that: unless the estimate is far above the line BOTH kernels are
    // launched and each checks the batch's exact row count on the device (m_max / m_min): one runs, the other exits at once.
    GemmDesc dd = d;
    if constexpr (LAYER >= 2) if (o.narrow_rows > 0 && d.N % 32 == 0) {
        const int lim = o.narrow_rows * (LAYER == 2 ? 1 : LAYER == 3 ? 2 : 4);
        const int est = rows_typ > 0 ? rows_typ : rows_hint;
        if (est <= 16 * lim) {
            dd.m_max = lim * d.rows_per_sample;
            const int rows_cov = rows_hint < lim ? rows_hint : lim;
            const dim3 grid((unsigned)skinny_waves_max(rows_cov * d.rows_per_sample, d.N)), block(64);
            const int nk = d.K / GBK;
            if (nk % 8 == 0) hipLaunchKernelGGL((k_gemm_skinny<LAYER, 8>), grid, block, 0, s, dd);
            else if (nk % 6 == 0) hipLaunchKernelGGL((k_gemm_skinny<LAYER, 6>), grid, block, 0, s, dd);
            else hipLaunchKernelGGL((k_gemm_skinny<LAYER, 2>), grid, block, 0, s, dd);
            if (rows_hint <= lim) return;                 // the bound itself is small: nothing else can be needed
            dd.m_min = dd.m_max;
        }
    }
    const GemmDesc& d2 = dd;
    if constexpr (LAYER == 1) if (d.N % HBN_ == 0 && d.cin % 64 == 0) {      // conv2 as a GEMM, image-resident (needs 256-channel multiples)
        if (d.states) launch_conv2_gemm<true>(d2, rows_hint, s);
        else launch_conv2_gemm<false>(d2, rows_hint, s);
        return;
    }
    if constexpr (LAYER == 2) {
        if (o.conv3_small && conv3_is_small(d2, rows_hint, rows_typ)) launch_ring_auto<LAYER>(d2, rows_hint, rows_typ, s, true);
        else launch_conv3_image(d2, rows_hint, s, o.conv3_tail != 0, o.conv3_planes != 0, o.conv3_pp);
        return;
    }

    launch_ring_auto<LAYER>(d2, rows_hint, rows_typ, s);
}

static hipEvent_t net_event(NetWorkspace* n) {
    if (!n->ev_pool.empty()) { hipEvent_t e = n->ev_pool.back(); n->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

bool netws_conv3_accounting(NetWorkspace* n, unsigned long long out[2], bool reset) {
    if (!n || !n->acct) return true;
    unsigned long long h[2] = {0, 0};
    if (hipMemcpy(h, n->acct, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return false;
    out[0] += h[0]; out[1] += h[1];
    return !reset || hipMemset(n->acct, 0, sizeof h) == hipSuccess;
}

// Resolve finished profile records (call after the workspace's stream is synchronised).
void netws_resolve_profile(NetWorkspace* n, NetProfile* prof) {
    if (!n) return;
    const int C = n->C;
    const double f_conv = 2.0 * 9.0 * C * C;
    const double per_sample = 2.0 * (42.0 * 18 * C) + f_conv * (42 + 20 + 6) + 2.0 * (6.0 * C * 1024 + 1024.0 * 512 + 512.0 * 8);
    if (!n->open.empty() && hipMemcpy(n->pinned_n, n->d_nlog, (size_t)n->pinned_next * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
        prof = nullptr;
    for (auto& r : n->open) {
        float conv2 = 0, conv3 = 0, conv4 = 0, fcs = 0, total = 0;
        if (hipEventElapsedTime(&conv2, r.e1, r.e2) == hipSuccess && hipEventElapsedTime(&conv3, r.e2, r.e2b) == hipSuccess &&
            hipEventElapsedTime(&conv4, r.e2b, r.e2c) == hipSuccess && hipEventElapsedTime(&fcs, r.e2c, r.e3) == hipSuccess &&
            hipEventElapsedTime(&total, r.e0, r.e3) == hipSuccess && prof) {
            const double rows = (double)*r.n;
            prof->conv2_ms += conv2;
            prof->conv3_ms += conv3;
            prof->conv3_flops += f_conv * 20.0 * rows;
            prof->conv4_ms += conv4;
            prof->conv4_flops += f_conv * 6.0 * rows;
            prof->fc_ms += fcs;
            prof->fc_flops += 2.0 * (6.0 * C * 1024 + 1024.0 * 512 + 512.0 * 8) * rows;
            prof->total_ms += total;
            prof->rows += rows;
            if (r.table2) {     // conv1 + conv2 are table lookups
                prof->conv2_bytes += rows * (304.0 * C * 2 + 42.0 * C * 2);      // 16 x 19 = 304 in-board (position, tap) pairs
                prof->total_flops += (per_sample - 2.0 * (42.0 * 18 * C) - f_conv * 42.0) * rows;
            } else {
                prof->conv2_flops += f_conv * 42.0 * rows;
                prof->total_flops += per_sample * rows;
            }
            prof->launches += 1;
        }
        n->ev_pool.push_back(r.e0); n->ev_pool.push_back(r.e1); n->ev_pool.push_back(r.e2); n->ev_pool.push_back(r.e2b); n->ev_pool.push_back(r.e2c);
        n->ev_pool.push_back(r.e3);
    }
    n->open.clear();
    n->pinned_next = 0;
}

bool netws_read_clock_stamps(NetWorkspace* n, unsigned long long* out2048) {
    return n && hipMemcpy(out2048, n->dbg, 2048 * 8, hipMemcpyDeviceToHost) == hipSuccess;
}

void convnet_forward(ConvNet* n, NetWorkspace* ws, const EvalBatch& eb, int rows_hint, int rows_typ, hipStream_t s, NetProfile* prof,
                     const NetOptions& o) {
    const int C = n->C;
    if (rows_hint > ws->max_batch) rows_hint = ws->max_batch;
    if (rows_hint <= 0) return;
    if (rows_typ <= 0 || rows_typ > rows_hint) rows_typ = rows_hint;
    NetWorkspace::Rec rec{};
    uint32_t* n_log = nullptr;
    const bool timed = prof != nullptr && ws->pinned_next < ws->pinned_cap;
    if (timed) {
        rec.e0 = net_event(ws); rec.e1 = net_event(ws); rec.e2 = net_event(ws); rec.e2b = net_event(ws); rec.e2c = net_event(ws); rec.e3 = net_event(ws);
        rec.n = ws->pinned_n + ws->pinned_next;
        n_log = ws->d_nlog + ws->pinned_next++;
        (void)hipEventRecord(rec.e0, s);
    }
    // conv1 + conv2: the default kernel set runs both as table gathers; conv2 as a GEMM gathers its image from the conv1 table
    // (256-channel multiples) or reads act1 written by k_conv1
    const bool shipped_set = o.gemm_variant == 5;
    const bool table2 = o.conv2_table && shipped_set;
    const bool table = !table2 && o.conv1_table && shipped_set && C % HBN_ == 0;
    rec.table2 = table2 ? 1 : 0;
    if (!table && !table2) {
        if (!netws_need_act1(ws)) return;
        const size_t waves = (size_t)rows_hint;                                        // one wave per board
        const size_t blocks = std::min<size_t>((waves + 3) / 4, 256 * 4);            // 4 persistent blocks per CU (38 KiB LDS each)
        hipLaunchKernelGGL(k_conv1, dim3((unsigned)blocks), dim3(256), (size_t)(19 * C) * sizeof(float), s, eb, n->w1, n->b1,
                           ws->act1, C);
    }
    GemmDesc d{};
    d.n_dev = eb.n;
    d.relu = 1;
    d.dbg = ws->dbg;
    d.c3tab = ws->c3tab;
    d.acct = ws->acct;
    // conv2: 3x3 same over the haloed [8][9][C] image (or the conv1 table) -> [6][7][C]
    d.A = table ? n->t1 : ws->act1; d.states = table ? eb.state : nullptr;
    d.W = n->wg[0]; d.bias = n->bg[0]; d.out = ws->act2;
    d.rows_per_sample = 42; d.out_w = 7; d.in_h = 8; d.in_w = 9; d.in_c = C; d.tap_w = 3; d.cin = C; d.K = 9 * C; d.N = C;
    if (timed) (void)hipEventRecord(rec.e1, s);
    if (table2) {
#ifdef AZ_DIAG
        if (o.conv2_table == 2) {
            const size_t blocks = std::min<size_t>(((size_t)rows_hint * 6 + 3) / 4, 256 * 8);   // one wave per (board, board row)
            hipLaunchKernelGGL(k_conv2_table, dim3((unsigned)blocks), dim3(256), (size_t)9 * C * 2, s, eb, n->u2, n->bg[0], ws->act2, C);
        } else
#endif
        {
            const size_t nsl = (size_t)C / 64;                                                  // one wave per (board, 64-channel slice)
            const size_t blocks = std::min<size_t>(((size_t)rows_hint + 3) / 4, 256) * nsl;
            hipLaunchKernelGGL(k_conv2_table_x, dim3((unsigned)blocks), dim3(256), 0, s, eb, n->u2, n->bg[0], ws->act2, C);
        }
    } else {
        launch_gemm<1>(d, rows_hint, rows_typ, s, o);
    }
    if (timed) (void)hipEventRecord(rec.e2, s);
    d.states = nullptr;
    // conv3: 3x3 valid [6][7][C] -> [4][5][C]
    d.A = ws->act2; d.W = n->wg[1]; d.bias = n->bg[1]; d.out = ws->act3;
    d.rows_per_sample = 20; d.out_w = 5; d.in_h = 6; d.in_w = 7;
    d.Wr = o.ring_packed ? n->wr[1] : nullptr;
    d.Wp = n->wp3;
    launch_gemm<2>(d, rows_hint, rows_typ, s, o);
    d.Wp = nullptr;
    if (timed) (void)hipEventRecord(rec.e2b, s);
    // conv4: 3x3 valid [4][5][C] -> [2][3][C]
    d.A = ws->act3; d.W = n->wg[2]; d.bias = n->bg[2]; d.out = ws->act4;
    d.Wr = o.ring_packed ? n->wr[2] : nullptr;
    d.rows_per_sample = 6; d.out_w = 3; d.in_h = 4; d.in_w = 5;
    launch_gemm<3>(d, rows_hint, rows_typ, s, o);
    if (timed) (void)hipEventRecord(rec.e2c, s);
    // fc1: [6C] -> 1024
    d.A = ws->act4; d.W = n->wg[3]; d.bias = n->bg[3]; d.out = ws->fc1o;
    d.Wr = o.ring_packed ? n->wr[3] : nullptr;
    d.rows_per_sample = 1; d.out_w = 1; d.in_h = 1; d.in_w = 1; d.in_c = 6 * C; d.tap_w = 1; d.cin = 6 * C; d.K = 6 * C; d.N = 1024;
    launch_gemm<4>(d, rows_hint, rows_typ, s, o);
    // fc2: 1024 -> 512
    d.A = ws->fc1o; d.W = n->wg[4]; d.bias = n->bg[4]; d.out = ws->fc2o;
    d.Wr = o.ring_packed ? n->wr[4] : nullptr;
    d.in_c = 1024; d.cin = 1024; d.K = 1024; d.N = 512;
    launch_gemm<5>(d, rows_hint, rows_typ, s, o);
    hipLaunchKernelGGL(k_heads, dim3((rows_hint * 64 + 255) / 256), dim3(256), 0, s, eb, ws->fc2o, n->wh, n->bh, n_log);
    if (timed) {
        (void)hipEventRecord(rec.e3, s);
        ws->open.push_back(rec);
    }
}

}  // namespace az
