// az_train.hip -- NNet::train (src/nnet.rs:38) as HIP kernels for gfx950: f32 state; by default the forward GEMMs of conv2..conv4 as
// f16 x 3 on the f16 matrix cores and the backward GEMMs as bf16 x 3 on the bf16 matrix cores (second half of this file: k_gemm3,
// k_gemm3_ring, k_wgrad3_tr, the gathered-operand forms), conv1 and the FC forward on the f32 matrix cores; "train_gemm" 0: every GEMM f32.
//
// Recipe (the reference's Python net, connect_four_net.py; only its hyper-parameters and layer list are taken,
// the TF1 code itself is broken -- SURVEY.md B11): loss = softmax cross-entropy(pi) + mean squared error(v)
// (:104-108), Adam lr 1e-3 (:21, :112), BatchNorm in training mode on every conv / FC (:39-77), dropout 0.3 on the
// two FC layers (:15, :72-89).  The data flow of one step:
//
//   boards -> col1 --GEMM--> z1 -BN,ReLU-> a1 -im2col-> col2 --GEMM--> z2 ... a4 = flat [b][6C]
//          --GEMM--> zf1 -BN,ReLU,dropout-> af1 --GEMM--> zf2 -BN,ReLU,dropout-> af2 -> (logits, v) -> loss
//   and back: heads -> BN/ReLU/dropout backward (two-stage column sums) -> wgrad GEMM (A^T dz), dgrad GEMM
//   (dz W^T) -> col2im -> previous layer ...; then one Adam kernel over the flat vector.
//
// Parameters, activations (row-major [rows][channels], rows = (sample, y, x)), gradients and the optimiser state are f32; k_gemm_f32
// (v_mfma_f32_16x16x4_f32, 64 x 64 or 128 x 128 block tiles, deterministic split-K) runs the forward GEMMs and, with "train_gemm" = 0,
// every GEMM.  At the reference's batch of 64 a step is ~63 GFLOP over 59 launches of 5 - 50 us.
#include "az_train.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "az_common.h"
#include "az_net.h"

namespace az {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- C[M][N] (+)= A(M x K) * B(K x N) (+ bias[N]) on the f32 matrix cores -------------------------------------
// A element (m,k) = A[m*sAm + k*sAk], B element (k,n) = B[k*sBk + n*sBn]: one of each pair of strides is 1, which
// selects the vector-load direction (A_K1: A contiguous along k; B_N1: B contiguous along n).  Tiles are staged
// k-major in LDS (As[k][m], Bs[k][n], row stride TM + 20 floats: the four k rows a wave reads sit 20 banks apart).
struct GemmF32 {
    const float* A; int64_t sAm, sAk;
    const float* B; int64_t sBk, sBn;
    float* C; int64_t ldc;  // splits == 1: the output; else partial sums [splits][M][N] (ldc = N)
    const float* bias;      // nullptr = none (splits == 1 only)
    int M, N, K;
    int k_per_split;        // multiple of TBK; blockIdx.z covers K range [z*k_per_split, +k_per_split)
    int splits;
};

constexpr int TBK = 16;

AZ_D float4 ld4_guard(const float* p, int valid) {   // up to 4 consecutive elements, zero-filled
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid >= 4 && (((uintptr_t)p) & 15) == 0) return *(const float4*)p;
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[1];
    if (valid > 2) v.z = p[2];
    if (valid > 3) v.w = p[3];
    return v;
}

// Block tile TM x TM (64 or 128), 4 waves in 2 x 2, each wave (TM/32)^2 MFMA tiles of 16 x 16.  The 128 tile halves the
// L2 -> CU bytes per flop (the three conv2-sized GEMMs of a step are stream-bound with 64 x 64 tiles); the 64 tile keeps
// the small GEMMs (FCs, conv4) spread over the chip.
// Split-K: a GEMM whose output has few tiles is cut along K into blockIdx.z slices so that every CU holds several
// blocks (that, not a deep software pipeline, is what hides the global-load latency of the short K loop); the slices
// are summed in a fixed order by k_splitk_reduce, so the result does not depend on scheduling.  The next tile's
// global loads are issued before the current tile's MFMAs.
template <int A_K1, int B_N1, int TM>
__global__ __launch_bounds__(256) void k_gemm_f32(const GemmF32 g) {
    constexpr int TLD = TM + 20;            // row stride of the k-major LDS tiles: 16-byte aligned rows, 20 banks apart
    constexpr int P = TM / 64;              // float4 loads per thread and operand
    constexpr int MT = TM / 32;             // 16 x 16 MFMA tiles per wave and dimension
    __shared__ __attribute__((aligned(16))) float As[TBK][TLD];
    __shared__ __attribute__((aligned(16))) float Bs[TBK][TLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TM;
    const int kbeg = blockIdx.z * g.k_per_split, kend = min(g.K, kbeg + g.k_per_split);
    f32x4 acc[MT][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;
    // this thread's slices of the A and B tiles: P x 4 consecutive elements along the contiguous direction.
    // contiguous along k (A_K1 / !B_N1): row (tid>>2) + 64*p, k quad (tid&3)*4;
    // contiguous along m/n: k row (tid / (TM/4)) + (1024/TM)*p, quad (tid % (TM/4))*4
    constexpr int QPR = TM / 4, KPP = 1024 / TM;
    const int a_r = A_K1 ? tid >> 2 : tid / QPR, a_q = A_K1 ? (tid & 3) * 4 : (tid % QPR) * 4;
    const int b_r = B_N1 ? tid / QPR : tid >> 2, b_q = B_N1 ? (tid % QPR) * 4 : (tid & 3) * 4;
    float4 va[P], vb[P];
#define AZ_TLOAD(k0_)                                                                                          \
    _Pragma("unroll") for (int p_ = 0; p_ < P; ++p_) {                                                         \
        va[p_] = make_float4(0.f, 0.f, 0.f, 0.f);                                                              \
        vb[p_] = make_float4(0.f, 0.f, 0.f, 0.f);                                                              \
        if (A_K1) { const int r_ = a_r + 64 * p_; if (m0 + r_ < g.M) va[p_] = ld4_guard(g.A + (int64_t)(m0 + r_) * g.sAm + ((k0_) + a_q), kend - ((k0_) + a_q)); } \
        else { const int r_ = a_r + KPP * p_; if ((k0_) + r_ < kend) va[p_] = ld4_guard(g.A + (int64_t)((k0_) + r_) * g.sAk + (m0 + a_q), g.M - (m0 + a_q)); }      \
        if (B_N1) { const int r_ = b_r + KPP * p_; if ((k0_) + r_ < kend) vb[p_] = ld4_guard(g.B + (int64_t)((k0_) + r_) * g.sBk + (n0 + b_q), g.N - (n0 + b_q)); } \
        else { const int r_ = b_r + 64 * p_; if (n0 + r_ < g.N) vb[p_] = ld4_guard(g.B + (int64_t)(n0 + r_) * g.sBn + ((k0_) + b_q), kend - ((k0_) + b_q)); }      \
    }
#define AZ_TSTORE()                                                                                            \
    _Pragma("unroll") for (int p_ = 0; p_ < P; ++p_) {                                                         \
        if (A_K1) { const int r_ = a_r + 64 * p_; As[a_q + 0][r_] = va[p_].x; As[a_q + 1][r_] = va[p_].y; As[a_q + 2][r_] = va[p_].z; As[a_q + 3][r_] = va[p_].w; } \
        else *(float4*)&As[a_r + KPP * p_][a_q] = va[p_];                                                      \
        if (B_N1) *(float4*)&Bs[b_r + KPP * p_][b_q] = vb[p_];                                                 \
        else { const int r_ = b_r + 64 * p_; Bs[b_q + 0][r_] = vb[p_].x; Bs[b_q + 1][r_] = vb[p_].y; Bs[b_q + 2][r_] = vb[p_].z; Bs[b_q + 3][r_] = vb[p_].w; } \
    }
    if (kbeg < kend) {
        AZ_TLOAD(kbeg);
        AZ_TSTORE();
    }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += TBK) {
        const bool more = k0 + TBK < kend;
        if (more) AZ_TLOAD(k0 + TBK);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = kk * 4 + fk;
            float a[MT], b[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) { a[i] = As[k][wm * (TM / 2) + i * 16 + fr]; b[i] = Bs[k][wn * (TM / 2) + i * 16 + fr]; }
            // the B tile is the instruction's first operand: D[n][m], a lane holds 4 consecutive n of one row m
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) AZ_TSTORE();
        __syncthreads();
    }
#undef AZ_TLOAD
#undef AZ_TSTORE
    float* cbase = g.splits > 1 ? g.C + (int64_t)blockIdx.z * g.M * g.N : g.C;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * (TM / 2) + i * 16 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int n = n0 + wn * (TM / 2) + j * 16 + fk * 4;
            float* c = cbase + (int64_t)m * g.ldc + n;
            if (n + 3 < g.N && ((((uintptr_t)c) & 15) == 0)) {
                float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                if (g.bias) { const float4 bv = *(const float4*)(g.bias + n); o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w; }
                *(float4*)c = o;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n + q < g.N) c[q] = acc[i][j][q] + (g.bias ? g.bias[n + q] : 0.0f);
            }
        }
    }
}

// ---- the forward GEMMs' kernel: C = A W (+ bias), A [M][K] and W [K][N] row-major, fed by LDS-DMA -----------------------------------
// k_gemm_f32 stages its tiles through registers (global load -> VGPR -> ds_write), a path a CU takes in at ~10 B/clk: conv2's forward GEMM
// reaches 83 of the 157 TFLOP/s of v_mfma_f32_16x16x4_f32 on it.  Here both tiles of a 16-deep K-step go global -> LDS by asm-issued
// global_load_lds_dwordx4 (no VGPR staging, no ds_write) into a ring of four 16 KiB stages, three in flight, retired by a counted
// s_waitcnt and ONE barrier per K-step; two workgroups per CU.  The LDS image is the global layout, 16-byte cells permuted on the source
// side so that the fragment reads are conflict-free:
//   A tile [128 m][16 k]: 64-byte rows of four k quads; lane (row fr, k group fk) reads quad fk of its row as ONE ds_read_b128 and uses
//     component j in the step's j-th MFMA -- so an MFMA sums k = j, 4 + j, 8 + j, 12 + j, and B follows the same assignment; quad q of
//     row r sits at position q ^ F(r), F = {0, 2, 3, 1}[(r >> 2) & 3] (the four non-contiguous 16-lane groups of a ds_read_b128 then
//     cover 16 distinct 16-byte slots);
//   B tile [16 k][128 n]: 512-byte rows; lane (n = fr, fk) reads B[4 fk + j][n] with ds_read_b32; 16-byte cell c of row k sits at
//     c ^ 4 ((k >> 2) & 1), which puts the two k groups of a 32-lane half on opposite halves of the banks.
// Requires K % 16 == 0, N % 128 == 0, 16-byte-aligned rows (every forward layer but conv1, whose K is 18).  The per-element summation
// order differs from k_gemm_f32's, so the last bits of z do too; tests/test_train_gpu.py holds both to float64 autograd.
constexpr int GD_BK = 16, GD_STAGE = 16384, GD_NS = 4;
AZ_D void gd_dma16(const void* sbase /*uniform*/, uint32_t voff, uint32_t lds_addr /*uniform*/) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__global__ __launch_bounds__(256, 2) void k_gemm_f32_dma(const GemmF32 g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[GD_NS * GD_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // workgroup ids go round the 8 XCDs: XCD x takes the x-th eighth of the list of (K slice, row tile, column tile), so the column tiles
    // that stream the same A rows share one XCD's L2 (k_gemm3_ring has the same map)
    const int NT = g.N / 128, tiles = ((g.M + 127) / 128) * NT, total = tiles * g.splits;
    const int nper = (total + 7) >> 3;
    const int t = ((int)blockIdx.x & 7) * nper + ((int)blockIdx.x >> 3);
    if (t >= total) return;
    const int split = t / tiles, rem = t - split * tiles;
    const int m0 = (rem / NT) * 128, n0 = (rem % NT) * 128;
    const int kbeg = split * g.k_per_split, kend = min(g.K, kbeg + g.k_per_split);
    const int nk = (kend - kbeg) / GD_BK;
    // DMA maps.  A: piece p (16 rows) of wave w = w, w + 4; lane -> row p*16 + (lane >> 2), position lane & 3 holding quad (lane & 3) ^ F(row)
    const uint32_t fA = (0x78u >> ((((uint32_t)lane >> 4) & 3u) * 2u)) & 3u;
    uint32_t a_ob[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = m0 + (wave + 4 * i) * 16 + (lane >> 2);
        m = m < g.M ? m : g.M - 1;
        a_ob[i] = (uint32_t)((int64_t)m * g.sAm * 4 + (((lane & 3) ^ fA) << 4));
    }
    // B: piece p (2 rows) of wave w = w, w + 4 (+ 8 rows: a uniform offset); lane -> row 2p + (lane >> 5), position lane & 31 holding cell
    // (lane & 31) ^ 4 ((row >> 2) & 1), and (row >> 2) & 1 == wave >> 1 for both pieces
    const uint32_t b_ob = (uint32_t)((int64_t)(2 * wave + (lane >> 5)) * g.sBk * 4 + (n0 + (((lane & 31) ^ ((wave >> 1) << 2)) << 2)) * 4);
    const size_t b_piece = (size_t)8 * g.sBk * 4;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)(smem + wave * 1024);
    int kd = kbeg;
#define AZ_GDDMA(buf_)                                                                                  \
    {                                                                                                   \
        const char* ab = (const char*)(g.A + kd);                                                       \
        const char* bb = (const char*)(g.B + (int64_t)kd * g.sBk);                                      \
        const uint32_t la = lds0 + (buf_) * GD_STAGE;                                                   \
        gd_dma16(ab, a_ob[0], la);                                                                      \
        gd_dma16(ab, a_ob[1], la + 4096);                                                               \
        gd_dma16(bb, b_ob, la + 8192);                                                                  \
        gd_dma16(bb + b_piece, b_ob, la + 8192 + 4096);                                                 \
        kd += GD_BK;                                                                                    \
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;
    const int a_off = (wr * 64 + fr) * 64 + ((fk ^ (int)((0x78u >> ((((uint32_t)fr >> 2) & 3u) * 2u)) & 3u)) << 4);
    int b_off[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) b_off[nt] = 8192 + fk * 4 * 512 + (((wc * 16 + nt * 4 + (fr >> 2)) ^ ((fk & 1) << 2)) << 4) + (fr & 3) * 4;
#pragma unroll
    for (int st = 0; st < GD_NS - 1; ++st)
        if (st < nk) AZ_GDDMA(st);
    for (int kt = 0; kt < nk; ++kt) {
        const int younger = nk - 1 - kt < GD_NS - 2 ? nk - 1 - kt : GD_NS - 2;      // stages issued after stage kt and still in flight
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // stage kt is complete for every wave; every wave is done with stage kt-1's buffer
        __builtin_amdgcn_sched_barrier(0);
        if (kt + GD_NS - 1 < nk) AZ_GDDMA((kt + GD_NS - 1) % GD_NS);
        const unsigned char* sb = smem + (kt % GD_NS) * GD_STAGE;
        float4 a4[4];
        float bv[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a4[mt] = *(const float4*)(sb + a_off + mt * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bv[nt][j] = *(const float*)(sb + b_off[nt] + j * 512);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float av = j == 0 ? a4[mt].x : j == 1 ? a4[mt].y : j == 2 ? a4[mt].z : a4[mt].w;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[nt][j], av, acc[mt][nt], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef AZ_GDDMA
    float* cbase = g.splits > 1 ? g.C + (int64_t)split * g.M * g.N : g.C;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wr * 64 + mt * 16 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wc * 64 + nt * 16 + fk * 4;
            float4 o = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
            if (g.bias) { const float4 bb = *(const float4*)(g.bias + n); o.x += bb.x; o.y += bb.y; o.z += bb.z; o.w += bb.w; }
            *(float4*)(cbase + (int64_t)m * g.ldc + n) = o;
        }
    }
}

// C[m][n] = sum over slices (in slice order) of partial[z][m][n] (+ bias[n])
__global__ void k_splitk_reduce(const float* __restrict__ partial, int splits, int M, int N, float* __restrict__ C, int64_t ldc,
                                const float* __restrict__ bias) {
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const int64_t m = i / N;
        float s = partial[i];
#pragma unroll 8
        for (int z = 1; z < splits; ++z) s += partial[(int64_t)z * total + i];
        C[m * ldc + n] = s + (bias ? bias[n] : 0.0f);
    }
}
// the same four columns at a time (N % 4 == 0, ldc % 4 == 0, 16-byte-aligned bases): the slices' loads of an element go out together
// (unrolled; the sum keeps its slice order), 16 bytes per lane
__global__ void k_splitk_reduce4(const float* __restrict__ partial, int splits, int M, int N, float* __restrict__ C, int64_t ldc,
                                 const float* __restrict__ bias) {
    const int64_t total4 = (int64_t)M * N / 4;
    const int n4 = N / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % n4) * 4;
        const int64_t m = i / n4;
        float4 s = *(const float4*)(partial + i * 4);
#pragma unroll 8
        for (int z = 1; z < splits; ++z) {
            const float4 v = *(const float4*)(partial + ((int64_t)z * total4 + i) * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (bias) { const float4 b = *(const float4*)(bias + n); s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w; }
        *(float4*)(C + m * ldc + n) = s;
    }
}
void launch_splitk_reduce(const float* partial, int splits, int M, int N, float* C, int64_t ldc, const float* bias, hipStream_t s) {
    const bool v4 = N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)partial & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0);
    if (v4) {
        const int64_t n = (int64_t)M * N / 4;
        hipLaunchKernelGGL(k_splitk_reduce4, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, partial, splits, M, N, C, ldc, bias);
    } else {
        const int64_t n = (int64_t)M * N;
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, partial, splits, M, N, C, ldc, bias);
    }
}


// ===================================================================================================================
// bf16 x 3: the f32 GEMMs of a step on the bf16 matrix cores ("train_gemm" = 1, the default)
// ===================================================================================================================
// v_mfma_f32_16x16x4_f32 peaks at 157 TFLOP/s and the register-staged kernel above reaches about half of it (its tiles come in through
// the ~10 B/clk/CU register-load path): 0.85 of the 1.58 ms step.  The bf16 matrix cores are 16 times faster per product, so every f32
// operand x is split into two bf16 numbers, hi = bf16(x) and lo = bf16(x - hi) (x = hi + lo up to 2^-17 |x|), and
//     A W  ~=  A_hi W_hi + A_hi W_lo + A_lo W_hi            (the dropped A_lo W_lo term is 2^-16 of a product)
// is three bf16 products per K-step accumulated in f32 by the MFMA.  k_gemm3 is built like the inference side's LDS-DMA kernels (asm-issued
// global_load_lds_dwordx4 into a swizzled image, two stages, one raw barrier per K-step) with all four operand tiles of a K-step in one
// stage, split-K over blockIdx.y and an f32 epilogue; k_splitk_reduce sums the slices in slice order, so a step stays deterministic.
// Both operands must be contiguous along the contraction:
//     dgrad     dA = dz W^T    dz [M][N] (written split by k_bn_bwd_apply), W [K][N] as stored (k_split_weights)
//     wgrad     dW = A^T dz    A^T [K][M'] and dz^T [N][M'] (k_transpose_split: f32 in, hi / lo out; M' = M rounded up to 64, zero-filled)
// THE FORWARD GEMMS ARE NOT bf16 x 3: a 2^-17 error in a pre-activation flips the ReLU (and the dropout-free BatchNorm sign) of the
// elements that lie that close to zero -- about one of fc2's 32 k activations per step and a few dozen per conv layer -- and ONE flip in an
// FC layer moves every upstream gradient tensor by ~3e-4 (tools/train_check.py: the forward as bf16 x 3 measured 3e-3 against float64
// autograd at batch 64, the f32 forward 1e-6).  The backward pass has no such discontinuity: its GEMMs' 1e-5 stays 1e-5.  Round 3 kept the
// forward on the f32 kernel; since round 4 conv2..conv4 run it as f16 x 3 (split_f16 below: 2^-22, the f32 kernel's own grade, with
// the operands scaled into half's NORMAL range -- the matrix cores treat half subnormals as zero).  conv1 (K = 18) and the FC layers
// (64 rows) stay on the f32 kernel.  Measured error of a step's gradients against float64 autograd: DESIGN.md section 8.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

AZ_D uint16_t bf16_rne(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
AZ_D float bf16_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
AZ_D void split_bf16(float x, uint16_t& hi, uint16_t& lo) {
    hi = bf16_rne(x);
    lo = bf16_rne(x - bf16_f32(hi));
}
// the same split in IEEE half precision (11-bit significands: hi + lo carries 22 bits).  For operands that stay inside half's range -- the
// forward pass's activations and its weights times 256 -- three products are f32-grade (2^-22), where bf16 x 3 is 2^-17.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
AZ_D void split_f16(float x, uint16_t& hi, uint16_t& lo) {
    const _Float16 h = (_Float16)x;
    hi = __builtin_bit_cast(uint16_t, h);
    lo = __builtin_bit_cast(uint16_t, (_Float16)(x - (float)h));
}
template <bool F16>
AZ_D void split_any(float x, uint16_t& hi, uint16_t& lo) {
    if constexpr (F16) split_f16(x, hi, lo);
    else split_bf16(x, hi, lo);
}
template <bool F16>
AZ_D f32x4 mfma3(const uint4 w, const uint4 a, const f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, a), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, a), c, 0, 0, 0);
}
AZ_D void split4(const float4 v, uint2& hi, uint2& lo) {
    uint16_t h0, h1, h2, h3, l0, l1, l2, l3;
    split_bf16(v.x, h0, l0); split_bf16(v.y, h1, l1); split_bf16(v.z, h2, l2); split_bf16(v.w, h3, l3);
    hi = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16));
    lo = make_uint2((uint32_t)l0 | ((uint32_t)l1 << 16), (uint32_t)l2 | ((uint32_t)l3 << 16));
}

// LDS-DMA from inline asm (az_net.hip lds_dma16: the compiler must not see the access, or every wait it inserts becomes vmcnt(0))
AZ_D void train_dma16(const void* sbase /*uniform*/, uint32_t voff, uint32_t lds_addr /*uniform*/) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

// A rows gathered from an activation tensor instead of read from an im2col matrix (k_gemm3_ring, k_wgrad3_tr): row m of the matrix is
// the position (s, y, x) of an Ht x Wt map, its K index is (tap, channel), and element (m, tap, c) is channel c of the source tensor
// [s][Hs][Ws][Cs] at (y + sgn * dy + off, x + sgn * dx + off), tap = 3 dy + dx -- or of the all-zero row at byte offset zero_off when that
// position is outside the source map.  Forward and wgrad: rows = output positions, source = the layer's input, sgn = +1, off = -pad;
// dgrad: rows = input positions, source = dz over the output positions, sgn = -1, off = +pad (the transposed convolution).  A 32-deep K-step
// never straddles a tap (Cs % 32 == 0).  The im2col matrix (9 x the tensor) is never written, and the tensor's rows come from L2.
struct ImplicitA {
    int on;
    int Ht, Wt, Hs, Ws, Cs;
    int sgn, off;
    uint32_t zero_off;
};

struct Gemm3 {
    const uint16_t *a_hi, *a_lo;    // [M][lda] bf16, contiguous along the contraction
    const uint16_t *w_hi, *w_lo;    // [N][ldw] bf16, contiguous along the contraction
    float* out;                     // splits == 1: [M][ldo] (+ bias); else partial sums [splits][M][N]
    const float* bias;
    int M, N, Kc;                   // Kc % 64 == 0, N % 128 == 0
    int lda, ldw, ldo;
    int steps_per_split;            // 32-deep K-steps (of Kc / 32) per blockIdx.y
    int splits;
    int xcd_rows;                   // workgroup id -> tile mapping (k_gemm3)
    float out_scale;                // the accumulators times this (a power of two: the f16 forward's weights are stored times 256), then + bias
    ImplicitA ia;                   // k_gemm3_ring only
};

// One stage = the four operand tiles of one 32-deep K-step -- A_hi, A_lo, W_hi, W_lo, 128 rows x 64 B each = 32 KiB -- shared by the step's
// three products (48 MFMAs per wave); two stages, two workgroups per CU.  (Walking the products as three K segments re-loaded A_hi and
// W_hi: 48 KiB for the same MFMAs, and the kernel is bound by that L2 -> LDS traffic.)  Rows of 64 bytes are four 16-byte k chunks; chunk q
// of row r sits at position q ^ F(r), F = {0, 2, 3, 1}[(r >> 2) & 3], which makes the ds_read_b128 fragment reads conflict-free
// (k_gemm_f32_dma's A tile has the same geometry).
constexpr int G3_BM = 128, G3_TILE = 128 * 64, G3_STAGE = 4 * G3_TILE;

template <bool F16>
__global__ __launch_bounds__(256, 2) void k_gemm3(const Gemm3 g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G3_STAGE];
    constexpr int MT = G3_BM / 32;
    const int NT = g.N / 128;
    const int id = blockIdx.x;
    int ntile, mtile;
    if (g.xcd_rows) {            // row tile = XCD + 8 i: the column tiles of a row tile share an XCD's L2 (workgroup ids go round the 8 XCDs)
        const int xcd = id & 7, j = id >> 3;
        ntile = j % NT; mtile = (j / NT) * 8 + xcd;
    } else {                     // fewer than 8 row tiles: that mapping would leave whole XCDs without work
        ntile = id % NT; mtile = id / NT;
    }
    const int m0 = mtile * G3_BM, n0 = ntile * 128;
    if (m0 >= g.M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // DMA map: a tile is 8 pieces of 16 rows; wave w loads pieces w and w + 4; lane -> row lane >> 2, position lane & 3 holding chunk
    // (lane & 3) ^ F(row)
    const uint32_t fD = (0x78u >> ((((uint32_t)lane >> 4) & 3u) * 2u)) & 3u;
    const uint32_t chunk = ((uint32_t)lane & 3u) ^ fD;
    uint32_t a_ob[2], b_ob[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = m0 + (wave + 4 * i) * 16 + (lane >> 2);
        m = m < g.M ? m : g.M - 1;
        a_ob[i] = (uint32_t)(m * g.lda + (int)chunk * 8) * 2u;
        b_ob[i] = (uint32_t)((n0 + (wave + 4 * i) * 16 + (lane >> 2)) * g.ldw + (int)chunk * 8) * 2u;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)(smem + wave * 1024);
    const int t0 = blockIdx.y * g.steps_per_split;
    const int nk = min(g.Kc / 32, t0 + g.steps_per_split) - t0;
    int kin = t0;
#define AZ_G3DMA(buf_)                                                                                  \
    {                                                                                                   \
        const char* ah = (const char*)(g.a_hi + kin * 32);                                              \
        const char* al = (const char*)(g.a_lo + kin * 32);                                              \
        const char* wh = (const char*)(g.w_hi + kin * 32);                                              \
        const char* wl = (const char*)(g.w_lo + kin * 32);                                              \
        const uint32_t la = lds0 + (buf_) * G3_STAGE;                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                              \
            train_dma16(ah, a_ob[i_], la + i_ * 4096);                                                  \
            train_dma16(al, a_ob[i_], la + G3_TILE + i_ * 4096);                                        \
            train_dma16(wh, b_ob[i_], la + 2 * G3_TILE + i_ * 4096);                                    \
            train_dma16(wl, b_ob[i_], la + 3 * G3_TILE + i_ * 4096);                                    \
        }                                                                                               \
        ++kin;                                                                                          \
    }
    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    const int coff = (fq ^ (int)((0x78u >> ((((uint32_t)frow >> 2) & 3u) * 2u)) & 3u)) << 4;
    const int a_row = (wr * (G3_BM / 2) + frow) * 64 + coff, b_row = (wc * 64 + frow) * 64 + coff;
    if (nk > 0) AZ_G3DMA(0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // stage kt has landed (the only one in flight)
        __builtin_amdgcn_s_barrier();                        // ... for every wave, and every wave is done with stage kt-1's buffer
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) AZ_G3DMA((kt + 1) & 1);
        const unsigned char* sAh = smem + (kt & 1) * G3_STAGE;
        const unsigned char* sAl = sAh + G3_TILE;
        const unsigned char* sWh = sAh + 2 * G3_TILE;
        const unsigned char* sWl = sAh + 3 * G3_TILE;
        uint4 ah[MT], al[MT], wh[4], wl[4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ah[mt] = *(const uint4*)(sAh + a_row + mt * 1024);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wh[nt] = *(const uint4*)(sWh + b_row + nt * 1024);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wl[nt] = *(const uint4*)(sWl + b_row + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) al[mt] = *(const uint4*)(sAl + a_row + mt * 1024);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma3<F16>(wh[nt], ah[mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma3<F16>(wl[nt], ah[mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma3<F16>(wh[nt], al[mt], acc[mt][nt]);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef AZ_G3DMA
    // f32 epilogue: a lane holds 4 consecutive columns of one row
    float* obase = g.splits > 1 ? g.out + (size_t)blockIdx.y * g.M * g.N : g.out;
    const int ldo = g.splits > 1 ? g.N : g.ldo;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wc * 64 + nt * 16 + fq * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias && g.splits == 1) bv = *(const float4*)(g.bias + n);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + wr * (G3_BM / 2) + mt * 16 + frow;
            if (m >= g.M) continue;
            *(float4*)(obase + (size_t)m * ldo + n) =
                make_float4(acc[mt][nt][0] * g.out_scale + bv.x, acc[mt][nt][1] * g.out_scale + bv.y, acc[mt][nt][2] * g.out_scale + bv.z,
                            acc[mt][nt][3] * g.out_scale + bv.w);
        }
    }
}

// The same product on ONE 8-wave workgroup per CU: tile 256 x 128, so a K-step moves 48 KiB (A hi / lo 16 KiB each, W hi / lo 8 KiB each)
// for the MFMAs k_gemm3's two workgroups need 64 KiB for, and a ring of THREE stages keeps two of them in flight all the time (k_gemm3
// has one, issued when the previous one has landed: its L2 -> LDS stream idles a latency per step and the kernel runs at about 11 B/clk
// per CU where the inference side's rings reach 19-25).  Wave w = (row group w >> 1 of 64 rows, column half w & 1): the same 64 x 64
// register tile, fragment geometry, chunk permutation and product order as k_gemm3 -- for one split-K plan the two kernels give the same
// bits.  DMA pieces of 16 rows (1 KiB): wave w loads pieces w and w + 8 of A hi and of A lo, piece w of W hi and of W lo (6 per stage).
constexpr int G3R_BM = 256, G3R_A = 256 * 64, G3R_W = 128 * 64, G3R_STAGE = 2 * G3R_A + 2 * G3R_W, G3R_NS = 3;

template <bool F16>
__global__ __launch_bounds__(512, 1) void k_gemm3_ring(const Gemm3 g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem_r[G3R_NS * G3R_STAGE];
    constexpr int MT = 4;
    // Workgroup ids go round the 8 XCDs: XCD x takes the x-th eighth of the list of (K slice, row tile, column tile) in that order, so the
    // column tiles of a row tile -- which stream the same A rows -- run on one XCD at the same time and share its L2, and every XCD gets
    // the same number of tiles whatever the tile counts are.
    const int NT = g.N / 128, tiles = ((g.M + G3R_BM - 1) / G3R_BM) * NT, total = tiles * g.splits;
    const int nper = (total + 7) >> 3;
    const int t = ((int)blockIdx.x & 7) * nper + ((int)blockIdx.x >> 3);
    if (t >= total) return;
    const int split = t / tiles, rem = t - split * tiles;
    const int mtile = rem / NT, ntile = rem - mtile * NT;
    const int m0 = mtile * G3R_BM, n0 = ntile * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const uint32_t fD = (0x78u >> ((((uint32_t)lane >> 4) & 3u) * 2u)) & 3u;
    const uint32_t chunk = ((uint32_t)lane & 3u) ^ fD;
    uint32_t a_ob[2], b_ob;
    int rs[2], ry[2], rx[2];          // implicit A: this lane's two rows as (sample, y, x)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = m0 + (wave + 8 * i) * 16 + (lane >> 2);
        m = m < g.M ? m : g.M - 1;
        a_ob[i] = (uint32_t)(m * g.lda + (int)chunk * 8) * 2u;
        rs[i] = ry[i] = rx[i] = 0;
        if (g.ia.on) {
            const int hw = g.ia.Ht * g.ia.Wt, p = m % hw;
            rs[i] = m / hw; ry[i] = p / g.ia.Wt; rx[i] = p - ry[i] * g.ia.Wt;
        }
    }
    b_ob = (uint32_t)((n0 + wave * 16 + (lane >> 2)) * g.ldw + (int)chunk * 8) * 2u;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)(smem_r + wave * 1024);
    const int t0 = split * g.steps_per_split;
    const int nk = min(g.Kc / 32, t0 + g.steps_per_split) - t0;
    int kin = t0;
    int tap = 0, tc = 0;              // implicit A: the tap and first channel of K-step kin
    if (g.ia.on) { tap = (t0 * 32) / g.ia.Cs; tc = t0 * 32 - tap * g.ia.Cs; }
#define AZ_G3RDMA(buf_)                                                                                 \
    {                                                                                                   \
        const char* ah = (const char*)(g.a_hi + kin * 32);                                              \
        const char* al = (const char*)(g.a_lo + kin * 32);                                              \
        const char* wh = (const char*)(g.w_hi + kin * 32);                                              \
        const char* wl = (const char*)(g.w_lo + kin * 32);                                              \
        uint32_t ao0 = a_ob[0], ao1 = a_ob[1];                                                          \
        if (g.ia.on) {                                                                                  \
            const int dy_ = tap / 3, dx_ = tap - 3 * dy_;                                               \
            const int sy0 = ry[0] + g.ia.sgn * dy_ + g.ia.off, sx0 = rx[0] + g.ia.sgn * dx_ + g.ia.off; \
            const int sy1 = ry[1] + g.ia.sgn * dy_ + g.ia.off, sx1 = rx[1] + g.ia.sgn * dx_ + g.ia.off; \
            const uint32_t cb = (uint32_t)(tc + (int)chunk * 8) * 2u;                                   \
            ao0 = ((unsigned)sy0 < (unsigned)g.ia.Hs && (unsigned)sx0 < (unsigned)g.ia.Ws)              \
                      ? (uint32_t)(((rs[0] * g.ia.Hs + sy0) * g.ia.Ws + sx0) * g.ia.Cs) * 2u + cb       \
                      : g.ia.zero_off + chunk * 16u;                                                    \
            ao1 = ((unsigned)sy1 < (unsigned)g.ia.Hs && (unsigned)sx1 < (unsigned)g.ia.Ws)              \
                      ? (uint32_t)(((rs[1] * g.ia.Hs + sy1) * g.ia.Ws + sx1) * g.ia.Cs) * 2u + cb       \
                      : g.ia.zero_off + chunk * 16u;                                                    \
            ah = (const char*)g.a_hi;                                                                   \
            al = (const char*)g.a_lo;                                                                   \
            tc += 32;                                                                                   \
            if (tc == g.ia.Cs) { tc = 0; ++tap; }                                                       \
        }                                                                                               \
        const uint32_t la = lds0 + (buf_) * G3R_STAGE;                                                  \
        train_dma16(ah, ao0, la);                                                                       \
        train_dma16(ah, ao1, la + 8192);                                                                \
        train_dma16(wh, b_ob, la + 2 * G3R_A);                                                          \
        train_dma16(wl, b_ob, la + 2 * G3R_A + G3R_W);                                                  \
        train_dma16(al, ao0, la + G3R_A);                                                               \
        train_dma16(al, ao1, la + G3R_A + 8192);                                                        \
        ++kin;                                                                                          \
    }
    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    const int coff = (fq ^ (int)((0x78u >> ((((uint32_t)frow >> 2) & 3u) * 2u)) & 3u)) << 4;
    const int a_row = (wr * 64 + frow) * 64 + coff, b_row = (wc * 64 + frow) * 64 + coff;
    if (nk > 0) AZ_G3RDMA(0);
    if (nk > 1) AZ_G3RDMA(1);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed: of this wave's pieces only stage kt + 1's six may still be in flight
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // ... for every wave, and every wave is done with stage kt - 1's buffer
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) { const int nb = buf == 0 ? 2 : buf - 1; AZ_G3RDMA(nb); }      // (kt + 2) % 3 == (kt - 1) % 3
        const unsigned char* sAh = smem_r + buf * G3R_STAGE;
        const unsigned char* sAl = sAh + G3R_A;
        const unsigned char* sWh = sAh + 2 * G3R_A;
        const unsigned char* sWl = sWh + G3R_W;
        uint4 ah[MT], al[MT], wh[4], wl[4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ah[mt] = *(const uint4*)(sAh + a_row + mt * 1024);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wh[nt] = *(const uint4*)(sWh + b_row + nt * 1024);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wl[nt] = *(const uint4*)(sWl + b_row + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) al[mt] = *(const uint4*)(sAl + a_row + mt * 1024);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma3<F16>(wh[nt], ah[mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma3<F16>(wl[nt], ah[mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma3<F16>(wh[nt], al[mt], acc[mt][nt]);
        __builtin_amdgcn_sched_barrier(0);
        buf = buf == 2 ? 0 : buf + 1;
    }
#undef AZ_G3RDMA
    float* obase = g.splits > 1 ? g.out + (size_t)split * g.M * g.N : g.out;
    const int ldo = g.splits > 1 ? g.N : g.ldo;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wc * 64 + nt * 16 + fq * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias && g.splits == 1) bv = *(const float4*)(g.bias + n);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + wr * 64 + mt * 16 + frow;
            if (m >= g.M) continue;
            *(float4*)(obase + (size_t)m * ldo + n) =
                make_float4(acc[mt][nt][0] * g.out_scale + bv.x, acc[mt][nt][1] * g.out_scale + bv.y, acc[mt][nt][2] * g.out_scale + bv.z,
                            acc[mt][nt][3] * g.out_scale + bv.w);
        }
    }
}

// wgrad without the transposes: dW [Kin][N] = A^T dz with A [M][Kin] and dz [M][N] AS STORED (the contraction index m is the row, the
// operand's own index is contiguous).  The MFMA operand of lane (i = lane & 15, g = lane >> 4) is 8 contraction values of index i, which
// gfx950 reads straight out of such an image: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4-row x 16-column block,
// two of them per fragment (rows 8g .. 8g+3, 8g+4 .. 8g+7).  k_transpose_split's pass over the layer's im2col matrix (32 us of a conv2-sized
// step, 76 us per step in all) is gone; the matrices come as bf16 hi / lo from k_im2col and k_bn_bwd_apply.
// Same tile and ring as k_gemm3_ring: 256 (Kin) x 128 (N), 8 waves as 4 x 2, a K-step = 32 rows of A (512-byte rows) and of dz (256-byte
// rows), hi and lo, 48 KiB; three stages.  The DMA is lane-linear, so the bank spread is made on the source side: the 32-byte unit u of
// row r sits at u ^ h(r), h(r) = (r & 3) | ((r >> 3) & 1) << 2 -- the eight rows a 32-lane half reads at once (4 per group, the groups 8
// rows apart) land on eight different units of the 256-byte bank period.
// Requires M % 32 == 0 (rows past M would have to read as zero), Kin % 256 == 0, N % 128 == 0.
struct Wgrad3 {
    const uint16_t *a_hi, *a_lo;    // [M][lda] bf16
    const uint16_t *z_hi, *z_lo;    // [M][ldz] bf16
    float* out;                     // splits == 1: dW [Kin][N]; else partial sums [splits][Kin][N]
    int Kin, N, M, lda, ldz;
    int steps_per_split, splits;    // 32-row K-steps per slice
    ImplicitA ia;                   // on: A is the im2col matrix of a_hi / a_lo = the layer's input tensor [s][Hs][Ws][Cs], Kin = 9 Cs, Cs % 256 == 0
    int inv_hw, inv_w;              // ceil(65536 / (Ht Wt)), ceil(65536 / Wt): the row walk's divisions as multiply-shifts
};
typedef short v4s_t __attribute__((ext_vector_type(4)));
AZ_D uint2 lds_tr16(const unsigned char* p) {
    typedef __attribute__((address_space(3))) v4s_t* lp;
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p));
}

__global__ __launch_bounds__(512, 1) void k_wgrad3_tr(const Wgrad3 g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem_t[G3R_NS * G3R_STAGE];
    constexpr int MT = 4;
    const int NT = g.N / 128, tiles = (g.Kin / 256) * NT, total = tiles * g.splits;
    const int nper = (total + 7) >> 3;
    const int t = ((int)blockIdx.x & 7) * nper + ((int)blockIdx.x >> 3);      // XCD x takes the x-th eighth of (slice, row tile, column tile)
    if (t >= total) return;
    const int split = t / tiles, rem = t - split * tiles;
    const int k0 = (rem / NT) * 256, n0 = (rem % NT) * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // DMA maps (a piece = 1 KiB of LDS = 64 lanes x 16 B): A piece p = rows 2p, 2p+1 (32 chunks each), dz piece p = rows 4p .. 4p+3 (16 chunks)
    uint32_t a_ob[2], z_ob;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 2 * (wave + 8 * i) + (lane >> 5);
        const int h = (r & 3) | (((r >> 3) & 1) << 2);
        a_ob[i] = (uint32_t)(r * g.lda + k0) * 2u + (uint32_t)(((lane & 31) ^ (h << 1)) << 4);
    }
    {
        const int r = 4 * wave + (lane >> 4);
        const int h = (r & 3) | (((r >> 3) & 1) << 2);
        z_ob = (uint32_t)(r * g.ldz + n0) * 2u + (uint32_t)(((lane & 15) ^ (h << 1)) << 4);
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)(smem_t + wave * 1024);
    const int t0 = split * g.steps_per_split;
    const int nk = min(g.M / 32, t0 + g.steps_per_split) - t0;
    int kin = t0;
    // implicit A: the tile's 256 columns lie inside one tap (Cs % 256 == 0); this lane's two rows of K-step kin as (sample, position)
    int ws_[2] = {0, 0}, wp_[2] = {0, 0};
    uint32_t icb[2] = {0u, 0u};
    int idy = 0, idx = 0;
    if (g.ia.on) {
        const int tap = k0 / g.ia.Cs, c0 = k0 - tap * g.ia.Cs, hw = g.ia.Ht * g.ia.Wt;
        idy = tap / 3; idx = tap - 3 * idy;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 2 * (wave + 8 * i) + (lane >> 5), m = t0 * 32 + r;
            const int h = (r & 3) | (((r >> 3) & 1) << 2);
            ws_[i] = m / hw; wp_[i] = m - ws_[i] * hw;
            icb[i] = (uint32_t)c0 * 2u + (uint32_t)(((lane & 31) ^ (h << 1)) << 4);
        }
    }
#define AZ_WTDMA(buf_)                                                                                  \
    {                                                                                                   \
        const char* ah = (const char*)(g.a_hi + (size_t)kin * 32 * g.lda);                              \
        const char* al = (const char*)(g.a_lo + (size_t)kin * 32 * g.lda);                              \
        const char* zh = (const char*)(g.z_hi + (size_t)kin * 32 * g.ldz);                              \
        const char* zl = (const char*)(g.z_lo + (size_t)kin * 32 * g.ldz);                              \
        uint32_t ao_[2] = {a_ob[0], a_ob[1]};                                                          \
        if (g.ia.on) {                                                                                  \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                          \
                const int y_ = (wp_[i_] * g.inv_w) >> 16, x_ = wp_[i_] - y_ * g.ia.Wt;                  \
                const int sy_ = y_ + idy + g.ia.off, sx_ = x_ + idx + g.ia.off;                         \
                ao_[i_] = ((unsigned)sy_ < (unsigned)g.ia.Hs && (unsigned)sx_ < (unsigned)g.ia.Ws)      \
                              ? (uint32_t)(((ws_[i_] * g.ia.Hs + sy_) * g.ia.Ws + sx_) * g.ia.Cs) * 2u + icb[i_] \
                              : g.ia.zero_off + (icb[i_] & 0x1F0u);                                     \
                const int p_ = wp_[i_] + 32, q_ = (p_ * g.inv_hw) >> 16;      /* the next K-step's rows are 32 further */ \
                ws_[i_] += q_;                                                                          \
                wp_[i_] = p_ - q_ * g.ia.Ht * g.ia.Wt;                                                  \
            }                                                                                           \
            ah = (const char*)g.a_hi;                                                                   \
            al = (const char*)g.a_lo;                                                                   \
        }                                                                                               \
        const uint32_t la = lds0 + (buf_) * G3R_STAGE;                                                  \
        train_dma16(ah, ao_[0], la);                                                                    \
        train_dma16(ah, ao_[1], la + 8192);                                                             \
        train_dma16(zh, z_ob, la + 2 * G3R_A);                                                          \
        train_dma16(zl, z_ob, la + 2 * G3R_A + G3R_W);                                                  \
        train_dma16(al, ao_[0], la + G3R_A);                                                            \
        train_dma16(al, ao_[1], la + G3R_A + 8192);                                                     \
        ++kin;                                                                                          \
    }
    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    // transposed-read addresses: lane 4q + p of group gq supplies row 8 gq + q (+ 4 for the second read), columns 4p .. 4p+3 of the block
    const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int hh = q | ((gq & 1) << 2);
    int a_off[MT], z_off[4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_off[mt] = (8 * gq + q) * 512 + (((wr * 4 + mt) ^ hh) << 5) + 8 * pp;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) z_off[nt] = (8 * gq + q) * 256 + (((wc * 4 + nt) ^ hh) << 5) + 8 * pp;
    if (nk > 0) AZ_WTDMA(0);
    if (nk > 1) AZ_WTDMA(1);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) { const int nb = buf == 0 ? 2 : buf - 1; AZ_WTDMA(nb); }
        const unsigned char* sAh = smem_t + buf * G3R_STAGE;
        const unsigned char* sAl = sAh + G3R_A;
        const unsigned char* sZh = sAh + 2 * G3R_A;
        const unsigned char* sZl = sZh + G3R_W;
        uint4 ah[MT], al[MT], zh[4], zl[4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const uint2 x = lds_tr16(sAh + a_off[mt]), y = lds_tr16(sAh + a_off[mt] + 4 * 512);
            ah[mt] = make_uint4(x.x, x.y, y.x, y.y);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const uint2 x = lds_tr16(sZh + z_off[nt]), y = lds_tr16(sZh + z_off[nt] + 4 * 256);
            zh[nt] = make_uint4(x.x, x.y, y.x, y.y);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const uint2 x = lds_tr16(sZl + z_off[nt]), y = lds_tr16(sZl + z_off[nt] + 4 * 256);
            zl[nt] = make_uint4(x.x, x.y, y.x, y.y);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const uint2 x = lds_tr16(sAl + a_off[mt]), y = lds_tr16(sAl + a_off[mt] + 4 * 512);
            al[mt] = make_uint4(x.x, x.y, y.x, y.y);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma3<false>(zh[nt], ah[mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma3<false>(zl[nt], ah[mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma3<false>(zh[nt], al[mt], acc[mt][nt]);
        __builtin_amdgcn_sched_barrier(0);
        buf = buf == 2 ? 0 : buf + 1;
    }
#undef AZ_WTDMA
    float* obase = g.splits > 1 ? g.out + (size_t)split * g.Kin * g.N : g.out;
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wc * 64 + nt * 16 + fq * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int k = k0 + wr * 64 + mt * 16 + frow;
            *(float4*)(obase + (size_t)k * g.N + n) = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
        }
    }
}

// every weight matrix W [K][N] f32 -> hi / lo bf16 as stored (the dgrad operand), at the matrix's own offset of flat buffers the size of
// the parameter vector.  blockIdx.y = matrix.
struct SplitWeights {
    int64_t off[5], count[5];
    int perm_c[5], perm_n[5];     // perm_c > 0: row (tap, c) of W [9 perm_c][perm_n] goes to row c, columns tap * perm_n .. of [perm_c][9 perm_n]
};
AZ_D void split_weights_body(const float* __restrict__ P, const SplitWeights& sw, uint16_t* __restrict__ w_hi, uint16_t* __restrict__ w_lo, int bx,
                             int mi, int gx) {
    const int64_t n4 = sw.count[mi] / 4, base = sw.off[mi];
    const int pc = sw.perm_c[mi], pn = sw.perm_n[mi];
    for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gx * blockDim.x) {
        uint2 hi, lo;
        split4(*(const float4*)(P + base + i * 4), hi, lo);
        int64_t o = i * 4;
        if (pc) {      // the dgrad of a convolution as a GEMM over (tap, n): W_d [c][tap][n] = W [tap][c][n]
            const int64_t row = o / pn;
            const int n = (int)(o - row * pn), tap = (int)(row / pc), c = (int)(row - (int64_t)tap * pc);
            o = ((int64_t)c * 9 + tap) * pn + n;
        }
        *(uint2*)(w_hi + base + o) = hi;
        *(uint2*)(w_lo + base + o) = lo;
    }
}
__global__ __launch_bounds__(256) void k_split_weights(const float* __restrict__ P, const SplitWeights sw, uint16_t* __restrict__ w_hi,
                                                       uint16_t* __restrict__ w_lo) {
    split_weights_body(P, sw, w_hi, w_lo, blockIdx.x, blockIdx.y, gridDim.x);
}

// in [R][C] f32 (row stride ld) -> out hi / lo [C][Rp] bf16, rows R .. Rp-1 of the source taken as zero; 64 x 64 tiles.  One launch
// transposes BOTH wgrad operands of a layer (dz and the layer's input): the tiles of the second follow the first's in blockIdx.x.
struct TransposeJob {
    const float* in;
    uint16_t *out_hi, *out_lo;
    int C, tiles_c;
    int64_t ld;
};
template <bool F16>
AZ_D void transpose_split_body(const TransposeJob& j0, const TransposeJob& j1, const TransposeJob& j2, int R, int Rp, float scale, int bx, int by,
                               float (*tile)[65]) {
    const int which = bx >= j0.tiles_c + j1.tiles_c ? 2 : (bx >= j0.tiles_c ? 1 : 0);
    const TransposeJob& j = which == 2 ? j2 : (which == 1 ? j1 : j0);
    const int r0 = by * 64, c0 = (bx - (which == 2 ? j0.tiles_c + j1.tiles_c : (which == 1 ? j0.tiles_c : 0))) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = ty + 4 * i;
        tile[r][tx] = (r0 + r < R && c0 + tx < j.C) ? j.in[(size_t)(r0 + r) * j.ld + c0 + tx] * scale : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = ty + 4 * i;
        if (c0 + c < j.C && r0 + tx < Rp) {
            uint16_t hi, lo;
            split_any<F16>(tile[tx][c], hi, lo);
            j.out_hi[(size_t)(c0 + c) * Rp + r0 + tx] = hi;
            j.out_lo[(size_t)(c0 + c) * Rp + r0 + tx] = lo;
        }
    }
}
template <bool F16>
__global__ __launch_bounds__(256) void k_transpose_split(const TransposeJob j0, const TransposeJob j1, const TransposeJob j2, int R, int Rp, float scale) {
    __shared__ float tile[64][65];
    transpose_split_body<F16>(j0, j1, j2, R, Rp, scale, blockIdx.x, blockIdx.y, tile);
}
// One launch for everything a step derives from the parameters: blocks [0, tx * ty) transpose conv2..conv4's matrices into the f16 x 3
// forward's operand ((256 W)^T as half pairs), the rest split the five matrices for dgrad (bf16 pairs, conv2's rows permuted).
__global__ __launch_bounds__(256) void k_weight_prep(const TransposeJob j0, const TransposeJob j1, const TransposeJob j2, int R, int Rp, float scale,
                                                     int tgx, int tgy, const float* __restrict__ P, const SplitWeights sw, uint16_t* __restrict__ w_hi,
                                                     uint16_t* __restrict__ w_lo, int sgx) {
    __shared__ float tile[64][65];
    const int id = blockIdx.x;
    if (id < tgx * tgy) { transpose_split_body<true>(j0, j1, j2, R, Rp, scale, id % tgx, id / tgx, tile); return; }
    const int r = id - tgx * tgy;
    split_weights_body(P, sw, w_hi, w_lo, r % sgx, r / sgx, sgx);
}

// what changes from step to step lives in device memory, so the launch sequence of a step is the same every time
struct StepState {
    uint64_t mask_seed;     // keys the dropout masks of this step
    float bc1, sqrt_bc2;    // Adam bias corrections 1 - beta1^t, sqrt(1 - beta2^t)
    int64_t idx_offset;     // first row of this step's batch in the epoch's index array
};

// epoch driver state: k_step_advance derives each step's StepState on the device
struct EpochCounters {
    uint64_t seed_key;      // mix64(train_seed ^ constant)
    uint64_t gstep;         // global step index (keys the dropout masks)
    int64_t epoch_step;     // step index inside the current epoch
    double beta1, beta2;
    double pow1, pow2;      // beta1^t, beta2^t as running products (the host keeps the same products: Trainer::pow1/2)
};
__global__ void k_step_advance(StepState* st, EpochCounters* c, int b) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st->mask_seed = mix64(c->seed_key ^ c->gstep);
    c->pow1 *= c->beta1;
    c->pow2 *= c->beta2;
    st->bc1 = (float)(1.0 - c->pow1);
    st->sqrt_bc2 = sqrtf((float)(1.0 - c->pow2));
    st->idx_offset = c->epoch_step * b;
    c->gstep += 1;
    c->epoch_step += 1;
}

// ---- data movement -------------------------------------------------------------------------------------------
// boards [b][2][6][7] planes -> conv1's im2col matrix col1 [b*42][20] (k = (ky*3+kx)*2 + ci, columns 18, 19 = 0)
__global__ void k_boards_col1(const float* __restrict__ boards, float* __restrict__ col, int b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * 42 * 20) return;
    const int k = i % 20, row = i / 20;
    const int s = row / 42, p = row % 42, y = p / 7, x = p % 7;
    float v = 0.0f;
    if (k < 18) {
        const int tap = k >> 1, ci = k & 1, iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
        if (iy >= 0 && iy < 6 && ix >= 0 && ix < 7) v = boards[(size_t)s * 84 + ci * 42 + iy * 7 + ix];
    }
    col[i] = v;
}

// in [b][H][W][C] -> col [b*Ho*Wo][9*C], k = (ky*3+kx)*C + c; pad = 1 ('same') or 0 ('valid')
// Every output is optional: col (f32: the f32 forward GEMM's and k_transpose_split's operand), col_hi / col_lo (64 x the matrix split into
// two halves, split_f16: the forward GEMM's operand when it runs as f16 x 3), col_bhi / col_blo (the matrix as bf16 hi / lo: k_wgrad3_tr's).
struct Im2colOut {
    float* col;
    uint16_t *col_hi, *col_lo, *col_bhi, *col_blo;
};
__global__ void k_im2col(const float* __restrict__ in, const Im2colOut out, int b, int H, int W, int C, int pad) {
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2, c4n = C / 4;
    const int64_t total = (int64_t)b * Ho * Wo * 9 * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int tap = (int)((i / c4n) % 9);
        const int64_t row = i / ((int64_t)c4n * 9);
        const int s = (int)(row / (Ho * Wo)), p = (int)(row % (Ho * Wo)), y = p / Wo, x = p % Wo;
        const int iy = y + tap / 3 - pad, ix = x + tap % 3 - pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const float4*)(in + (((size_t)s * H + iy) * W + ix) * C + c4 * 4);
        const size_t o = (size_t)row * 9 * C + (size_t)tap * C + c4 * 4;
        if (out.col) *(float4*)(out.col + o) = v;
        if (out.col_hi) {
            uint16_t h0, h1, h2, h3, l0, l1, l2, l3;
            // times 64: a half below 2^-14 is subnormal, and the low halves of activations under 1/8 would be; scaled, that is under 1/512
            split_f16(v.x * 64.0f, h0, l0); split_f16(v.y * 64.0f, h1, l1); split_f16(v.z * 64.0f, h2, l2); split_f16(v.w * 64.0f, h3, l3);
            *(uint2*)(out.col_hi + o) = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16));
            *(uint2*)(out.col_lo + o) = make_uint2((uint32_t)l0 | ((uint32_t)l1 << 16), (uint32_t)l2 | ((uint32_t)l3 << 16));
        }
        if (out.col_bhi) {
            uint2 hi, lo;
            split4(v, hi, lo);
            *(uint2*)(out.col_bhi + o) = hi;
            *(uint2*)(out.col_blo + o) = lo;
        }
    }
}

// transpose of k_im2col as a gather: din[s][iy][ix][c] = sum over taps of dcol[(s, iy-ky+pad, ix-kx+pad)][tap*C + c]
__global__ void k_col2im(const float* __restrict__ dcol, float* __restrict__ din, int b, int H, int W, int C, int pad) {
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2, c4n = C / 4;
    const int64_t total = (int64_t)b * H * W * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int64_t pos = i / c4n;
        const int s = (int)(pos / (H * W)), p = (int)(pos % (H * W)), iy = p / W, ix = p % W;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int oy = iy - tap / 3 + pad, ox = ix - tap % 3 + pad;
            if (oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
            const float4 v = *(const float4*)(dcol + (((size_t)s * Ho + oy) * Wo + ox) * 9 * C + (size_t)tap * C + c4 * 4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        *(float4*)(din + (size_t)pos * C + c4 * 4) = a;
    }
}

// The step's batch gathered from the epoch's samples (boards, pis, vs: rows idx[idx_offset ..]) and conv1's im2col matrix (k_boards_col1's)
// straight from those rows, one launch (the epoch loop)
__global__ void k_gather_col1(const float* __restrict__ all_boards, const float* __restrict__ all_pis, const float* __restrict__ all_vs,
                              const int64_t* __restrict__ idx, int b, float* __restrict__ boards, float* __restrict__ pis, float* __restrict__ vs,
                              float* __restrict__ col, const StepState* __restrict__ st) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t off = st->idx_offset;
    if (i < b * 92) {
        const int j = i / 92, f = i % 92;
        const int64_t src = idx[off + j];
        if (f < 84) boards[(size_t)j * 84 + f] = all_boards[(size_t)src * 84 + f];
        else if (f < 91) pis[(size_t)j * 7 + (f - 84)] = all_pis[(size_t)src * 7 + (f - 84)];
        else vs[j] = all_vs[src];
    }
    if (i < b * 42 * 20) {
        const int k = i % 20, row = i / 20;
        const int s = row / 42, p = row % 42, y = p / 7, x = p % 7;
        float v = 0.0f;
        if (k < 18) {
            const int tap = k >> 1, ci = k & 1, iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
            if (iy >= 0 && iy < 6 && ix >= 0 && ix < 7) v = all_boards[(size_t)idx[off + s] * 84 + ci * 42 + iy * 7 + ix];
        }
        col[i] = v;
    }
}

// ---- BatchNorm (training mode) + ReLU + dropout ----------------------------------------------------------------
struct BnLayer {
    const float* z;         // [M][N] pre-BN
    float* out;             // forward: a [M][N]; backward: dz [M][N]
    const float* grad_out;  // backward: d loss / d a [M][N]
    const float* gamma;
    const float* beta;
    float* mean;            // [N] batch mean
    float* invstd;          // [N]
    int M, N;
    uint32_t drop_layer;    // dropout stream id; keep_thresh = 0 -> no dropout
    uint32_t keep_thresh;
    float drop_scale;
    uint16_t *out_hi, *out_lo;   // k_bn_bwd_apply, not nullptr: dz also as hi / lo bf16 (the dgrad operand), same [M][N] layout
    // k_bn_apply, not nullptr: the activations also as 64 x a in half-precision hi / lo (the next layer's f16 x 3 forward operand) and as
    // bf16 hi / lo (its wgrad operand), same [M][N] layout -- what k_im2col writes 9 x of when the GEMMs do not gather (ImplicitA)
    uint16_t *act_hi, *act_lo, *act_bhi, *act_blo;
};


// d loss / d (BatchNorm output) of element (r, c) behind ReLU and dropout, and xhat
AZ_D float bn_grad_in(const BnLayer& L, uint64_t mask_seed, size_t i, float z, float go, float mean, float invstd, float gamma, float beta, float& xh) {
    xh = (z - mean) * invstd;
    const float y = gamma * xh + beta;
    float g = y > 0.0f ? go : 0.0f;
    if (L.keep_thresh) g = dropout_keep(mask_seed, L.drop_layer, i, L.keep_thresh) ? g * L.drop_scale : 0.0f;
    return g;
}

// Geometry of the four BatchNorm kernels: a block is 32 column quads (128 columns, one float4 per thread and row) x 8 row lanes; a
// thread walks its rows 8 apart with four rows' loads in flight.  (Round 2's blocks were 64 single columns x 4 row lanes with one
// 4-byte load in flight per thread: 12.5 us for the 11 MB a conv2-sized backward reduction reads, 0.9 TB/s; N % 128 == 0 for every layer.)
constexpr int BN_COLS = 128, BN_LANES = 8;

// stage 1 of a column reduction: the block reduces rows [blockIdx.y*rpb, +rpb) of its 128 columns and writes
// partial[blockIdx.y][column][2] (f64); KIND 0: (sum z, sum z^2); 1: BN backward (sum g, sum g*xhat).  The eight row lanes are
// combined in lane order, the slices by the consumers in slice order: no atomics, one result whatever the scheduling.
template <int KIND>
__global__ __launch_bounds__(256) void k_colreduce(const BnLayer L, int rpb, double* __restrict__ partial,
                                                   const StepState* __restrict__ st) {
    __shared__ double red[BN_LANES][32][8];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * BN_COLS + tx * 4;
    const int r0 = blockIdx.y * rpb, r1 = min(L.M, r0 + rpb);
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    float4 mean = make_float4(0, 0, 0, 0), invstd = mean, gamma = mean, beta = mean;
    uint64_t mask_seed = 0;
    if (KIND == 1) {
        mean = *(const float4*)(L.mean + c); invstd = *(const float4*)(L.invstd + c);
        gamma = *(const float4*)(L.gamma + c); beta = *(const float4*)(L.beta + c);
        if (L.keep_thresh) mask_seed = st->mask_seed;
    }
    // a thread's rows of the slice are requested together, twelve at a time (a slice of the conv layers has 64 .. 84 rows = 8 .. 11 per thread):
    // one round trip per slice instead of one per four rows
    auto accum = [&](size_t i, const float4 z, const float4 go) {
        if (KIND == 0) {
            s0[0] += z.x; s1[0] += (double)z.x * z.x; s0[1] += z.y; s1[1] += (double)z.y * z.y;
            s0[2] += z.z; s1[2] += (double)z.z * z.z; s0[3] += z.w; s1[3] += (double)z.w * z.w;
        } else {
            float xh;
            float g = bn_grad_in(L, mask_seed, i + 0, z.x, go.x, mean.x, invstd.x, gamma.x, beta.x, xh); s0[0] += g; s1[0] += (double)g * xh;
            g = bn_grad_in(L, mask_seed, i + 1, z.y, go.y, mean.y, invstd.y, gamma.y, beta.y, xh); s0[1] += g; s1[1] += (double)g * xh;
            g = bn_grad_in(L, mask_seed, i + 2, z.z, go.z, mean.z, invstd.z, gamma.z, beta.z, xh); s0[2] += g; s1[2] += (double)g * xh;
            g = bn_grad_in(L, mask_seed, i + 3, z.w, go.w, mean.w, invstd.w, gamma.w, beta.w, xh); s0[3] += g; s1[3] += (double)g * xh;
        }
    };
    constexpr int PER = 12;
    for (int rb = r0 + ty; rb < r1; rb += PER * BN_LANES) {
        float4 zv[PER], gv[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int r = rb + q * BN_LANES;
            const size_t i = (size_t)r * L.N + c;
            zv[q] = r < r1 ? *(const float4*)(L.z + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (KIND == 1) gv[q] = r < r1 ? *(const float4*)(L.grad_out + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int r = rb + q * BN_LANES;
            if (r < r1) accum((size_t)r * L.N + c, zv[q], KIND == 1 ? gv[q] : zv[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[ty][tx][2 * q] = s0[q]; red[ty][tx][2 * q + 1] = s1[q]; }
    __syncthreads();
    // 256 threads = 128 columns x 2 values: sum the eight row lanes in lane order
    const int cl = threadIdx.x >> 1, v = threadIdx.x & 1;
    double t = 0.0;
#pragma unroll
    for (int l = 0; l < BN_LANES; ++l) t += red[l][cl >> 2][2 * (cl & 3) + v];
    partial[((size_t)blockIdx.y * L.N + blockIdx.x * BN_COLS + cl) * 2 + v] = t;
}

// Stage 2 is folded into the consumers: every block of the apply kernels first sums the partials of its 128 columns in slice
// order (<= 32 slices: cheap, and the same result in every block), so there is no separate finish launch.
AZ_D void bn_sum_partials(const double* __restrict__ partial, int nparts, int N, int col0, double (*sums)[2]) {
    const int cl = threadIdx.x >> 1, v = threadIdx.x & 1;
    double t = 0.0;
#pragma unroll 8
    for (int p = 0; p < nparts; ++p) t += partial[((size_t)p * N + col0 + cl) * 2 + v];      // unrolled: the loads go out together, the sum keeps its order
    sums[cl][v] = t;
}

// a = dropout(relu(gamma * xhat + beta)).  grid (N/128, row blocks of `rpb` rows).
// Batch mean / biased variance -> mean, invstd (kept for the backward pass); the blockIdx.y == 0 blocks also update
// the moving averages in place (moving = momentum*moving + (1-momentum)*batch, the variance with Bessel's
// correction, as F.batch_norm does).
__global__ __launch_bounds__(256) void k_bn_apply(const BnLayer L, const double* __restrict__ partial, int nparts, int rpb, float eps,
                                                  float momentum, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                  const StepState* __restrict__ st) {
    __shared__ double sums[BN_COLS][2];
    __shared__ __attribute__((aligned(16))) float s_mean[BN_COLS], s_inv[BN_COLS];
    const int col0 = blockIdx.x * BN_COLS;
    // this thread's rows (rpb <= 32: at most four, 8 apart) are requested BEFORE the statistics are summed, so that their round trip
    // runs under that prologue instead of after it
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, c = col0 + tx * 4;
    const int r0 = blockIdx.y * rpb, r1 = min(L.M, r0 + rpb);
    float4 zpre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + ty + q * BN_LANES;
        zpre[q] = r < r1 ? *(const float4*)(L.z + (size_t)r * L.N + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bn_sum_partials(partial, nparts, L.N, col0, sums);
    __syncthreads();
    if (threadIdx.x < BN_COLS) {
        const int cl = threadIdx.x, c = col0 + cl;
        const double mu = sums[cl][0] / L.M;
        double var = sums[cl][1] / L.M - mu * mu;
        if (var < 0.0) var = 0.0;
        const float mean = (float)mu, inv = (float)(1.0 / sqrt(var + (double)eps));
        s_mean[cl] = mean; s_inv[cl] = inv;
        if (blockIdx.y == 0) {
            L.mean[c] = mean; L.invstd[c] = inv;
            const double unbiased = L.M > 1 ? var * L.M / (L.M - 1) : var;
            run_mean[c] = momentum * run_mean[c] + (1.0f - momentum) * (float)mu;
            run_var[c] = momentum * run_var[c] + (1.0f - momentum) * (float)unbiased;
        }
    }
    __syncthreads();
    const float4 mean = *(const float4*)(s_mean + tx * 4), inv = *(const float4*)(s_inv + tx * 4);
    const float4 gamma = *(const float4*)(L.gamma + c), beta = *(const float4*)(L.beta + c);
    const uint64_t mask_seed = L.keep_thresh ? st->mask_seed : 0;
    auto row = [&](int r, const float4 z) {
        const size_t i = (size_t)r * L.N + c;
        float4 y = make_float4(fmaxf(gamma.x * ((z.x - mean.x) * inv.x) + beta.x, 0.0f), fmaxf(gamma.y * ((z.y - mean.y) * inv.y) + beta.y, 0.0f),
                               fmaxf(gamma.z * ((z.z - mean.z) * inv.z) + beta.z, 0.0f), fmaxf(gamma.w * ((z.w - mean.w) * inv.w) + beta.w, 0.0f));
        if (L.keep_thresh) {
            y.x = dropout_keep(mask_seed, L.drop_layer, (uint64_t)i + 0, L.keep_thresh) ? y.x * L.drop_scale : 0.0f;
            y.y = dropout_keep(mask_seed, L.drop_layer, (uint64_t)i + 1, L.keep_thresh) ? y.y * L.drop_scale : 0.0f;
            y.z = dropout_keep(mask_seed, L.drop_layer, (uint64_t)i + 2, L.keep_thresh) ? y.z * L.drop_scale : 0.0f;
            y.w = dropout_keep(mask_seed, L.drop_layer, (uint64_t)i + 3, L.keep_thresh) ? y.w * L.drop_scale : 0.0f;
        }
        *(float4*)(L.out + i) = y;
        if (L.act_hi) {
            uint16_t h0, h1, h2, h3, l0, l1, l2, l3;
            split_f16(y.x * 64.0f, h0, l0); split_f16(y.y * 64.0f, h1, l1); split_f16(y.z * 64.0f, h2, l2); split_f16(y.w * 64.0f, h3, l3);
            *(uint2*)(L.act_hi + i) = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16));
            *(uint2*)(L.act_lo + i) = make_uint2((uint32_t)l0 | ((uint32_t)l1 << 16), (uint32_t)l2 | ((uint32_t)l3 << 16));
        }
        if (L.act_bhi) {
            uint2 bh, bl;
            split4(y, bh, bl);
            *(uint2*)(L.act_bhi + i) = bh;
            *(uint2*)(L.act_blo + i) = bl;
        }
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + ty + q * BN_LANES;
        if (r < r1) row(r, zpre[q]);
    }
    for (int r = r0 + ty + 4 * BN_LANES; r < r1; r += BN_LANES) row(r, *(const float4*)(L.z + (size_t)r * L.N + c));
}

// dz = gamma * invstd * (g - (dbeta + xhat * dgamma) / M), dgamma = sum g*xhat, dbeta = sum g (same grid as k_bn_apply)
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const BnLayer L, const double* __restrict__ partial, int nparts, int rpb,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      const StepState* __restrict__ st) {
    __shared__ double sums[BN_COLS][2];
    __shared__ __attribute__((aligned(16))) float s_db[BN_COLS], s_dg[BN_COLS];
    const int col0 = blockIdx.x * BN_COLS;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, c = col0 + tx * 4;
    const int r0 = blockIdx.y * rpb, r1 = min(L.M, r0 + rpb);
    float4 zpre[4], gpre[4];          // requested before the prologue (see k_bn_apply)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + ty + q * BN_LANES;
        const size_t i = (size_t)r * L.N + c;
        zpre[q] = r < r1 ? *(const float4*)(L.z + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        gpre[q] = r < r1 ? *(const float4*)(L.grad_out + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bn_sum_partials(partial, nparts, L.N, col0, sums);
    __syncthreads();
    if (threadIdx.x < BN_COLS) {
        const int cl = threadIdx.x, c = col0 + cl;
        s_db[cl] = (float)sums[cl][0]; s_dg[cl] = (float)sums[cl][1];
        if (blockIdx.y == 0) { dbeta[c] = (float)sums[cl][0]; dgamma[c] = (float)sums[cl][1]; }
    }
    __syncthreads();
    const float4 mean = *(const float4*)(L.mean + c), invstd = *(const float4*)(L.invstd + c);
    const float4 gamma = *(const float4*)(L.gamma + c), beta = *(const float4*)(L.beta + c);
    const float4 db = *(const float4*)(s_db + tx * 4), dg = *(const float4*)(s_dg + tx * 4);
    const float inv_m = 1.0f / (float)L.M;
    const uint64_t mask_seed = L.keep_thresh ? st->mask_seed : 0;
    auto row = [&](int r, const float4 z, const float4 go) {
        const size_t i = (size_t)r * L.N + c;
        float xh;
        float4 dz;
        float g = bn_grad_in(L, mask_seed, i + 0, z.x, go.x, mean.x, invstd.x, gamma.x, beta.x, xh); dz.x = gamma.x * invstd.x * (g - (db.x + xh * dg.x) * inv_m);
        g = bn_grad_in(L, mask_seed, i + 1, z.y, go.y, mean.y, invstd.y, gamma.y, beta.y, xh); dz.y = gamma.y * invstd.y * (g - (db.y + xh * dg.y) * inv_m);
        g = bn_grad_in(L, mask_seed, i + 2, z.z, go.z, mean.z, invstd.z, gamma.z, beta.z, xh); dz.z = gamma.z * invstd.z * (g - (db.z + xh * dg.z) * inv_m);
        g = bn_grad_in(L, mask_seed, i + 3, z.w, go.w, mean.w, invstd.w, gamma.w, beta.w, xh); dz.w = gamma.w * invstd.w * (g - (db.w + xh * dg.w) * inv_m);
        *(float4*)(L.out + i) = dz;
        if (L.out_hi) {
            uint2 hi, lo;
            split4(dz, hi, lo);
            *(uint2*)(L.out_hi + i) = hi;
            *(uint2*)(L.out_lo + i) = lo;
        }
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + ty + q * BN_LANES;
        if (r < r1) row(r, zpre[q], gpre[q]);
    }
    for (int r = r0 + ty + 4 * BN_LANES; r < r1; r += BN_LANES) {
        const size_t i = (size_t)r * L.N + c;
        row(r, *(const float4*)(L.z + i), *(const float4*)(L.grad_out + i));
    }
}

// The FC layers' BatchNorm (at most BN_SMALL_ROWS rows: one slice): both stages in ONE launch -- a block owns 128 columns of EVERY row, so
// it has the complete column sums itself (same lane-order summation as k_colreduce with one slice: identical results).
constexpr int BN_SMALL_ROWS = 64;
// A thread's rows (at most BN_SMALL_ROWS / BN_LANES = 8, 8 apart) are loaded ONCE, all in flight together, and kept in registers for
// both stages: the kernels were four dependent round trips (two per stage), now one.
constexpr int BN_SMALL_PER = BN_SMALL_ROWS / BN_LANES;
template <int KIND>
AZ_D void bn_small_load(const BnLayer& L, float4 (&zr)[BN_SMALL_PER], float4 (&gr)[BN_SMALL_PER]) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * BN_COLS + tx * 4;
#pragma unroll
    for (int q = 0; q < BN_SMALL_PER; ++q) {
        const int r = ty + q * BN_LANES;
        const size_t i = (size_t)r * L.N + c;
        zr[q] = r < L.M ? *(const float4*)(L.z + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (KIND == 1) gr[q] = r < L.M ? *(const float4*)(L.grad_out + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int KIND>
AZ_D void bn_small_sums(const BnLayer& L, uint64_t mask_seed, float4 mean, float4 invstd, float4 gamma, float4 beta, double (*red)[32][8],
                        double (*sums)[2], const float4 (&zr)[BN_SMALL_PER], const float4 (&gr)[BN_SMALL_PER]) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * BN_COLS + tx * 4;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < BN_SMALL_PER; ++q) {
        const int r = ty + q * BN_LANES;
        if (r >= L.M) break;
        const size_t i = (size_t)r * L.N + c;
        const float4 z = zr[q];
        if (KIND == 0) {
            s0[0] += z.x; s1[0] += (double)z.x * z.x; s0[1] += z.y; s1[1] += (double)z.y * z.y;
            s0[2] += z.z; s1[2] += (double)z.z * z.z; s0[3] += z.w; s1[3] += (double)z.w * z.w;
        } else {
            const float4 go = gr[q];
            float xh;
            float g = bn_grad_in(L, mask_seed, i + 0, z.x, go.x, mean.x, invstd.x, gamma.x, beta.x, xh); s0[0] += g; s1[0] += (double)g * xh;
            g = bn_grad_in(L, mask_seed, i + 1, z.y, go.y, mean.y, invstd.y, gamma.y, beta.y, xh); s0[1] += g; s1[1] += (double)g * xh;
            g = bn_grad_in(L, mask_seed, i + 2, z.z, go.z, mean.z, invstd.z, gamma.z, beta.z, xh); s0[2] += g; s1[2] += (double)g * xh;
            g = bn_grad_in(L, mask_seed, i + 3, z.w, go.w, mean.w, invstd.w, gamma.w, beta.w, xh); s0[3] += g; s1[3] += (double)g * xh;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[ty][tx][2 * q] = s0[q]; red[ty][tx][2 * q + 1] = s1[q]; }
    __syncthreads();
    const int cl = threadIdx.x >> 1, v = threadIdx.x & 1;
    double t = 0.0;
#pragma unroll
    for (int l = 0; l < BN_LANES; ++l) t += red[l][cl >> 2][2 * (cl & 3) + v];
    sums[cl][v] = t;
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_bn_fwd_small(const BnLayer L, float eps, float momentum, float* __restrict__ run_mean,
                                                      float* __restrict__ run_var, const StepState* __restrict__ st) {
    __shared__ double red[BN_LANES][32][8];
    __shared__ double sums[BN_COLS][2];
    __shared__ __attribute__((aligned(16))) float s_mean[BN_COLS], s_inv[BN_COLS];
    const float4 z4 = make_float4(0, 0, 0, 0);
    float4 zr[BN_SMALL_PER], gr[BN_SMALL_PER];
    bn_small_load<0>(L, zr, gr);
    bn_small_sums<0>(L, 0, z4, z4, z4, z4, red, sums, zr, gr);
    const int col0 = blockIdx.x * BN_COLS;
    if (threadIdx.x < BN_COLS) {
        const int cl = threadIdx.x, c = col0 + cl;
        const double mu = sums[cl][0] / L.M;
        double var = sums[cl][1] / L.M - mu * mu;
        if (var < 0.0) var = 0.0;
        const float mean = (float)mu, inv = (float)(1.0 / sqrt(var + (double)eps));
        s_mean[cl] = mean; s_inv[cl] = inv;
        L.mean[c] = mean; L.invstd[c] = inv;
        const double unbiased = L.M > 1 ? var * L.M / (L.M - 1) : var;
        run_mean[c] = momentum * run_mean[c] + (1.0f - momentum) * (float)mu;
        run_var[c] = momentum * run_var[c] + (1.0f - momentum) * (float)unbiased;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, c = col0 + tx * 4;
    const float4 mean = *(const float4*)(s_mean + tx * 4), inv = *(const float4*)(s_inv + tx * 4);
    const float4 gamma = *(const float4*)(L.gamma + c), beta = *(const float4*)(L.beta + c);
    const uint64_t mask_seed = L.keep_thresh ? st->mask_seed : 0;
#pragma unroll
    for (int q = 0; q < BN_SMALL_PER; ++q) {
        const int r = ty + q * BN_LANES;
        if (r >= L.M) break;
        const size_t i = (size_t)r * L.N + c;
        const float4 z = zr[q];
        float4 y = make_float4(fmaxf(gamma.x * ((z.x - mean.x) * inv.x) + beta.x, 0.0f), fmaxf(gamma.y * ((z.y - mean.y) * inv.y) + beta.y, 0.0f),
                               fmaxf(gamma.z * ((z.z - mean.z) * inv.z) + beta.z, 0.0f), fmaxf(gamma.w * ((z.w - mean.w) * inv.w) + beta.w, 0.0f));
        if (L.keep_thresh) {
            y.x = dropout_keep(mask_seed, L.drop_layer, (uint64_t)i + 0, L.keep_thresh) ? y.x * L.drop_scale : 0.0f;
            y.y = dropout_keep(mask_seed, L.drop_layer, (uint64_t)i + 1, L.keep_thresh) ? y.y * L.drop_scale : 0.0f;
            y.z = dropout_keep(mask_seed, L.drop_layer, (uint64_t)i + 2, L.keep_thresh) ? y.z * L.drop_scale : 0.0f;
            y.w = dropout_keep(mask_seed, L.drop_layer, (uint64_t)i + 3, L.keep_thresh) ? y.w * L.drop_scale : 0.0f;
        }
        *(float4*)(L.out + i) = y;
        if (L.act_bhi) {
            uint2 bh, bl;
            split4(y, bh, bl);
            *(uint2*)(L.act_bhi + i) = bh;
            *(uint2*)(L.act_blo + i) = bl;
        }
    }
}

__global__ __launch_bounds__(256) void k_bn_bwd_small(const BnLayer L, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      const StepState* __restrict__ st) {
    __shared__ double red[BN_LANES][32][8];
    __shared__ double sums[BN_COLS][2];
    const int col0 = blockIdx.x * BN_COLS;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, c = col0 + tx * 4;
    const float4 mean = *(const float4*)(L.mean + c), invstd = *(const float4*)(L.invstd + c);
    const float4 gamma = *(const float4*)(L.gamma + c), beta = *(const float4*)(L.beta + c);
    const uint64_t mask_seed = L.keep_thresh ? st->mask_seed : 0;
    float4 zr[BN_SMALL_PER], gr[BN_SMALL_PER];
    bn_small_load<1>(L, zr, gr);
    bn_small_sums<1>(L, mask_seed, mean, invstd, gamma, beta, red, sums, zr, gr);
    if (threadIdx.x < BN_COLS) { dbeta[col0 + threadIdx.x] = (float)sums[threadIdx.x][0]; dgamma[col0 + threadIdx.x] = (float)sums[threadIdx.x][1]; }
    const float4 db = make_float4((float)sums[tx * 4][0], (float)sums[tx * 4 + 1][0], (float)sums[tx * 4 + 2][0], (float)sums[tx * 4 + 3][0]);
    const float4 dg = make_float4((float)sums[tx * 4][1], (float)sums[tx * 4 + 1][1], (float)sums[tx * 4 + 2][1], (float)sums[tx * 4 + 3][1]);
    const float inv_m = 1.0f / (float)L.M;
#pragma unroll
    for (int q = 0; q < BN_SMALL_PER; ++q) {
        const int r = ty + q * BN_LANES;
        if (r >= L.M) break;
        const size_t i = (size_t)r * L.N + c;
        const float4 z = zr[q], go = gr[q];
        float xh;
        float4 dz;
        float g = bn_grad_in(L, mask_seed, i + 0, z.x, go.x, mean.x, invstd.x, gamma.x, beta.x, xh); dz.x = gamma.x * invstd.x * (g - (db.x + xh * dg.x) * inv_m);
        g = bn_grad_in(L, mask_seed, i + 1, z.y, go.y, mean.y, invstd.y, gamma.y, beta.y, xh); dz.y = gamma.y * invstd.y * (g - (db.y + xh * dg.y) * inv_m);
        g = bn_grad_in(L, mask_seed, i + 2, z.z, go.z, mean.z, invstd.z, gamma.z, beta.z, xh); dz.z = gamma.z * invstd.z * (g - (db.z + xh * dg.z) * inv_m);
        g = bn_grad_in(L, mask_seed, i + 3, z.w, go.w, mean.w, invstd.w, gamma.w, beta.w, xh); dz.w = gamma.w * invstd.w * (g - (db.w + xh * dg.w) * inv_m);
        *(float4*)(L.out + i) = dz;
        if (L.out_hi) {
            uint2 hi, lo;
            split4(dz, hi, lo);
            *(uint2*)(L.out_hi + i) = hi;
            *(uint2*)(L.out_lo + i) = lo;
        }
    }
}

// ---- heads, loss and their gradients -----------------------------------------------------------------------------
// one block (64 threads) per sample: logits = a pi_w + pi_b, v = tanh(a v_w + v_b);
// loss_pi_j = -sum_a pi_a log_softmax_a, loss_v_j = (v - t)^2; dhead[j][0..6] = (softmax * sum(pi) - pi) / b,
// dhead[j][7] = 2 (v - t) (1 - v^2) / b
__global__ __launch_bounds__(64) void k_heads_loss(const float* __restrict__ a, const float* __restrict__ pi_w,
                                                   const float* __restrict__ pi_b, const float* __restrict__ v_w,
                                                   const float* __restrict__ v_b, const float* __restrict__ tpi,
                                                   const float* __restrict__ tv, int b, float* __restrict__ dhead,
                                                   float* __restrict__ sample_loss, float* __restrict__ out_logits) {
    const int j = blockIdx.x, lane = threadIdx.x;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = lane; k < 512; k += 64) {
        const float x = a[(size_t)j * 512 + k];
#pragma unroll
        for (int o = 0; o < 7; ++o) acc[o] += x * pi_w[k * 7 + o];
        acc[7] += x * v_w[k];
    }
#pragma unroll
    for (int o = 0; o < 8; ++o)
        for (int off = 32; off > 0; off >>= 1) acc[o] += __shfl_xor(acc[o], off, 64);
    if (lane == 0) {
        float logit[7], mx = -INFINITY, tsum = 0.0f;
        for (int o = 0; o < 7; ++o) { logit[o] = acc[o] + pi_b[o]; mx = fmaxf(mx, logit[o]); tsum += tpi[(size_t)j * 7 + o]; }
        float se = 0.0f;
        for (int o = 0; o < 7; ++o) se += expf(logit[o] - mx);
        const float lse = mx + logf(se);
        float lp = 0.0f;
        for (int o = 0; o < 7; ++o) {
            const float t = tpi[(size_t)j * 7 + o], ls = logit[o] - lse;
            lp -= t * ls;
            dhead[(size_t)j * 8 + o] = (expf(ls) * tsum - t) / (float)b;
            if (out_logits) out_logits[(size_t)j * 8 + o] = logit[o];
        }
        const float v = tanhf(acc[7] + v_b[0]), dv = v - tv[j];
        dhead[(size_t)j * 8 + 7] = 2.0f * dv * (1.0f - v * v) / (float)b;
        if (out_logits) out_logits[(size_t)j * 8 + 7] = v;
        sample_loss[2 * j] = lp;
        sample_loss[2 * j + 1] = dv * dv;
    }
}

// d pi_w[k][o] = sum_j a[j][k] dhead[j][o], d v_w[k] = sum_j a[j][k] dhead[j][7]; biases = column sums of dhead;
// da[j][k] = sum_o dhead[j][o] pi_w[k][o] + dhead[j][7] v_w[k]
__global__ void k_heads_bwd(const float* __restrict__ a, const float* __restrict__ dhead, const float* __restrict__ pi_w,
                            const float* __restrict__ v_w, int b, float* __restrict__ d_pi_w, float* __restrict__ d_pi_b,
                            float* __restrict__ d_v_w, float* __restrict__ d_v_b, float* __restrict__ da,
                            const float* __restrict__ sample_loss, double* __restrict__ totals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {       // fixed-order sum of the per-sample losses into the running totals (b <= 256)
        double lp = 0.0, lv = 0.0;
#pragma unroll 16
        for (int j = 0; j < b; ++j) { lp += sample_loss[2 * j]; lv += sample_loss[2 * j + 1]; }      // unrolled: the loads go out together, the sum keeps its order
        totals[0] += lp / b;
        totals[1] += lv / b;
    }
    if (i < 512 * 8) {
        const int k = i >> 3, o = i & 7;
        float s = 0.0f;
#pragma unroll 16
        for (int j = 0; j < b; ++j) s += a[(size_t)j * 512 + k] * dhead[(size_t)j * 8 + o];
        if (o < 7) d_pi_w[k * 7 + o] = s; else d_v_w[k] = s;
    } else if (i < 512 * 8 + 8) {
        const int o = i - 512 * 8;
        float s = 0.0f;
#pragma unroll 16
        for (int j = 0; j < b; ++j) s += dhead[(size_t)j * 8 + o];
        if (o < 7) d_pi_b[o] = s; else d_v_b[0] = s;
    }
    for (int64_t e = i; e < (int64_t)b * 512; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e >> 9), k = (int)(e & 511);
        float s = dhead[(size_t)j * 8 + 7] * v_w[k];
#pragma unroll
        for (int o = 0; o < 7; ++o) s += dhead[(size_t)j * 8 + o] * pi_w[k * 7 + o];
        da[e] = s;
    }
}

// ---- Adam (torch.optim.Adam form: p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)) ------------------------------------
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       int64_t n, float lr, float b1, float b2, float eps, const StepState* __restrict__ st) {
    const float bc1 = st->bc1, sqrt_bc2 = st->sqrt_bc2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / sqrt_bc2 + eps);
    }
}

// ---- host side --------------------------------------------------------------------------------------------------
struct Trainer {
    int C = 512;
    Layout L{512};
    std::vector<void*> dev;
    float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;
    // batch scratch
    float *bboards = nullptr, *bpis = nullptr, *bvs = nullptr;
    // activations: col[l] (GEMM input of layer l), z[l], a[l] for the 4 convs and the 2 FCs
    float* col[4] = {nullptr};
    float *z[6] = {nullptr}, *a[6] = {nullptr};
    float *mean[6] = {nullptr}, *invstd[6] = {nullptr};
    float *dz = nullptr, *dact = nullptr, *dcol = nullptr;     // backward scratch (largest layer)
    float *sums = nullptr, *dhead = nullptr, *sample_loss = nullptr, *logits = nullptr;
    double *partial = nullptr, *loss_totals = nullptr;
    float* splitk = nullptr;           // partial sums of the split-K GEMMs
    // bf16 x 3 operands of the backward GEMMs (gemm_mode 1): the weights as stored, dz, and the transposed wgrad operands
    int gemm_mode = 1;
    bool fwd_dma = true;               // forward GEMMs on k_gemm_f32_dma where its shape constraints hold
    uint16_t *w_hi = nullptr, *w_lo = nullptr;                                         // [L.total] each, a matrix at its own offset
    uint16_t *dz_hi = nullptr, *dz_lo = nullptr, *dzt_hi = nullptr, *dzt_lo = nullptr, *at_hi = nullptr, *at_lo = nullptr;
    StepState* step_state = nullptr;
    StepState* host_state = nullptr;   // pinned ring: source of the asynchronous per-step uploads of trainer_step
    uint32_t host_state_next = 0;
    EpochCounters* counters = nullptr;
    bool use_graph = true;
    size_t splitk_floats = 0;
    // second branch of a step (gemm_mode 1): the weight split and every layer's wgrad chain (transposes, k_gemm3, split-K reduce) run on
    // `side`, forked from / joined to the caller's stream by events, with their own split-K workspace; the caller's stream keeps the
    // chain the next layer waits for (BatchNorm backward, dgrad, col2im).  Same kernels on the same data: bit-identical to fork = false.
    // forward conv2..conv4 as f16 x 3 (fwd_x3): the im2col matrices and the transposed weights (times 256) as half-precision hi / lo pairs
    bool fwd_x3 = true;
    uint16_t *col_hi[4] = {nullptr}, *col_lo[4] = {nullptr}, *wt_hi[4] = {nullptr}, *wt_lo[4] = {nullptr};
    // wgrad of conv2..conv4 on k_wgrad3_tr (transposed LDS reads, no k_transpose_split): the im2col matrices as bf16 hi / lo
    bool wgrad_tr = true;
    uint16_t *col_bhi[4] = {nullptr}, *col_blo[4] = {nullptr};
    // the three conv GEMMs of conv2..conv4 gather their A rows from the activations (ImplicitA): no im2col matrix, no col2im.  The
    // activations a[0..2] as half / bf16 pairs, each with one all-zero row behind the largest batch's rows (act_zero_off[l] bytes in)
    bool implicit = true;
    uint16_t *act_hi[3] = {nullptr}, *act_lo[3] = {nullptr}, *act_bhi[5] = {nullptr}, *act_blo[5] = {nullptr};      // [3], [4]: the FC layers' inputs (their wgrad)
    uint32_t act_zero_off[3] = {0, 0, 0}, dz_zero_off = 0;
    bool gemm3_ring = true;            // dgrad / wgrad with >= 192 rows on k_gemm3_ring (256 x 128 tiles, 3-stage ring) instead of k_gemm3
    bool fork = false;                 // measured: no gain as direct launches, 7 % slower inside a hipGraph (profiles/README.md)
    hipStream_t side = nullptr;
    hipEvent_t ev_start = nullptr, ev_sw = nullptr, ev_join = nullptr, ev_dz[6] = {nullptr}, ev_tr[6] = {nullptr};
    float* splitk2 = nullptr;
    size_t splitk2_floats = 0;
    int64_t step = 0;
    double pow1 = 1.0, pow2 = 1.0;     // beta1^step, beta2^step
    template <class T> T* dalloc(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        dev.push_back(p);
        return (T*)p;
    }
};

constexpr int RED_PARTS = 32;
constexpr uint32_t HOST_STATE_RING = 64;   // trainer_step's callers synchronise at least this often (az_net_train_step: every step)

Trainer* trainer_create(int channels, const char** err) {
    if (channels % 128 != 0 || channels < 128) { if (err) *err = "net_channels must be a multiple of 128"; return nullptr; }
    Trainer* t = new Trainer();
    t->C = channels;
    t->L = Layout(channels);
    const size_t B = TRAIN_MAX_BATCH, C = (size_t)channels, T = (size_t)t->L.total;
    bool ok = true;
    ok &= (t->params = t->dalloc<float>(T)) != nullptr;
    ok &= (t->grads = t->dalloc<float>(T)) != nullptr;
    ok &= (t->m = t->dalloc<float>(T)) != nullptr;
    ok &= (t->v = t->dalloc<float>(T)) != nullptr;
    ok &= (t->bboards = t->dalloc<float>(B * 84)) != nullptr;
    ok &= (t->bpis = t->dalloc<float>(B * 7)) != nullptr;
    ok &= (t->bvs = t->dalloc<float>(B)) != nullptr;
    const size_t rows[6] = {B * 42, B * 42, B * 20, B * 6, B, B};
    const size_t kin[4] = {20, 9 * C, 9 * C, 9 * C};
    const size_t nout[6] = {C, C, C, C, 1024, 512};
    for (int l = 0; l < 4; ++l) ok &= (t->col[l] = t->dalloc<float>(rows[l] * kin[l])) != nullptr;
    for (int l = 0; l < 6; ++l) {
        ok &= (t->z[l] = t->dalloc<float>(rows[l] * nout[l])) != nullptr;
        ok &= (t->a[l] = t->dalloc<float>(rows[l] * nout[l])) != nullptr;
        ok &= (t->mean[l] = t->dalloc<float>(nout[l])) != nullptr;
        ok &= (t->invstd[l] = t->dalloc<float>(nout[l])) != nullptr;
    }
    ok &= (t->dz = t->dalloc<float>(B * 42 * std::max<size_t>(C, 1024))) != nullptr;
    ok &= (t->dact = t->dalloc<float>(B * 42 * std::max<size_t>(C, 1024))) != nullptr;
    ok &= (t->dcol = t->dalloc<float>(B * 42 * 9 * C)) != nullptr;
    ok &= (t->sums = t->dalloc<float>(2 * std::max<size_t>(C, 1024))) != nullptr;
    ok &= (t->dhead = t->dalloc<float>(B * 8)) != nullptr;
    ok &= (t->logits = t->dalloc<float>(B * 8)) != nullptr;
    ok &= (t->sample_loss = t->dalloc<float>(B * 2)) != nullptr;
    ok &= (t->partial = t->dalloc<double>((size_t)RED_PARTS * std::max<size_t>(C, 1024) * 2)) != nullptr;
    ok &= (t->loss_totals = t->dalloc<double>(2)) != nullptr;
    ok &= (t->step_state = t->dalloc<StepState>(1)) != nullptr;
    ok &= hipHostMalloc((void**)&t->host_state, HOST_STATE_RING * sizeof(StepState), hipHostMallocDefault) == hipSuccess;
    ok &= (t->counters = t->dalloc<EpochCounters>(1)) != nullptr;
    {
        const size_t NM = std::max<size_t>(C, 1024), Mp = (B * 42 + 63) / 64 * 64;
        for (uint16_t** q : {&t->w_hi, &t->w_lo}) ok &= (*q = t->dalloc<uint16_t>(T)) != nullptr;
        for (uint16_t** q : {&t->dz_hi, &t->dz_lo}) {      // + one all-zero row (the gathered dgrad's out-of-map positions)
            ok &= (*q = t->dalloc<uint16_t>((B * 42 + 1) * NM)) != nullptr;
            if (*q) ok &= hipMemset(*q, 0, (B * 42 + 1) * NM * sizeof(uint16_t)) == hipSuccess;
        }
        t->dz_zero_off = (uint32_t)(B * 42 * NM * sizeof(uint16_t));
        for (uint16_t** q : {&t->dzt_hi, &t->dzt_lo}) ok &= (*q = t->dalloc<uint16_t>(NM * Mp)) != nullptr;
        for (uint16_t** q : {&t->at_hi, &t->at_lo}) ok &= (*q = t->dalloc<uint16_t>(9 * C * Mp)) != nullptr;
        for (int l = 0; l < 3; ++l) {
            const size_t n = (rows[l] + 1) * C;
            for (uint16_t** q : {&t->act_hi[l], &t->act_lo[l], &t->act_bhi[l], &t->act_blo[l]}) {
                ok &= (*q = t->dalloc<uint16_t>(n)) != nullptr;
                if (*q) ok &= hipMemset(*q, 0, n * sizeof(uint16_t)) == hipSuccess;
            }
            t->act_zero_off[l] = (uint32_t)(rows[l] * C * sizeof(uint16_t));
        }
        for (int l = 3; l < 5; ++l)
            for (uint16_t** q : {&t->act_bhi[l], &t->act_blo[l]}) ok &= (*q = t->dalloc<uint16_t>(rows[l] * nout[l])) != nullptr;
        for (int l = 1; l < 4; ++l) {
            for (uint16_t** q : {&t->col_hi[l], &t->col_lo[l], &t->col_bhi[l], &t->col_blo[l]}) ok &= (*q = t->dalloc<uint16_t>(rows[l] * kin[l])) != nullptr;
            for (uint16_t** q : {&t->wt_hi[l], &t->wt_lo[l]}) ok &= (*q = t->dalloc<uint16_t>(kin[l] * C)) != nullptr;
        }
    }
    t->splitk_floats = (size_t)96 << 20;      // 384 MiB (a conv2-sized dgrad at the largest batch has 49.5 M outputs; it is never split)
    ok &= (t->splitk = t->dalloc<float>(t->splitk_floats)) != nullptr;
    t->splitk2_floats = (size_t)16 << 20;     // wgrad outputs are [K][N] <= 1152 x 128 or 768 x 1024 floats per slice
    ok &= (t->splitk2 = t->dalloc<float>(t->splitk2_floats)) != nullptr;
    if (!ok) { if (err) *err = "hipMalloc failed for the trainer workspace"; trainer_destroy(t); return nullptr; }
    (void)hipMemset(t->loss_totals, 0, 2 * sizeof(double));
    return t;
}

void trainer_destroy(Trainer* t) {
    if (!t) return;
    if (t->side) { (void)hipStreamSynchronize(t->side); (void)hipStreamDestroy(t->side); }
    for (hipEvent_t ev : {t->ev_start, t->ev_sw, t->ev_join}) if (ev) (void)hipEventDestroy(ev);
    for (int l = 0; l < 6; ++l) { if (t->ev_dz[l]) (void)hipEventDestroy(t->ev_dz[l]); if (t->ev_tr[l]) (void)hipEventDestroy(t->ev_tr[l]); }
    for (void* p : t->dev) (void)hipFree(p);
    if (t->host_state) (void)hipHostFree(t->host_state);
    delete t;
}

bool trainer_set_params(Trainer* t, const float* host_params, int64_t count) {
    if (!t || count != t->L.total) return false;
    const size_t bytes = (size_t)count * sizeof(float);
    if (hipMemcpy(t->params, host_params, bytes, hipMemcpyHostToDevice) != hipSuccess) return false;
    if (hipMemset(t->m, 0, bytes) != hipSuccess || hipMemset(t->v, 0, bytes) != hipSuccess ||
        hipMemset(t->grads, 0, bytes) != hipSuccess || hipMemset(t->loss_totals, 0, 2 * sizeof(double)) != hipSuccess)
        return false;
    t->step = 0;
    t->pow1 = t->pow2 = 1.0;
    return true;
}

bool trainer_get_params(Trainer* t, float* host_params, int64_t count, hipStream_t s) {
    if (!t || count != t->L.total) return false;
    if (hipStreamSynchronize(s) != hipSuccess) return false;
    return hipMemcpy(host_params, t->params, (size_t)count * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
}

bool trainer_get_grads(Trainer* t, float* host_grads, int64_t count, hipStream_t s) {
    if (!t || count != t->L.total) return false;
    if (hipStreamSynchronize(s) != hipSuccess) return false;
    return hipMemcpy(host_grads, t->grads, (size_t)count * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
}

bool trainer_read_losses(Trainer* t, double out[2], bool reset, hipStream_t s) {
    if (!t) return false;
    if (hipStreamSynchronize(s) != hipSuccess) return false;
    if (hipMemcpy(out, t->loss_totals, 2 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return false;
    if (reset && hipMemset(t->loss_totals, 0, 2 * sizeof(double)) != hipSuccess) return false;
    return true;
}

float* trainer_batch_boards(Trainer* t) { return t->bboards; }
float* trainer_batch_pis(Trainer* t) { return t->bpis; }
float* trainer_batch_vs(Trainer* t) { return t->bvs; }

namespace {

inline dim3 grid1(int64_t n, int block = 256, int cap = 4096) { return dim3((unsigned)std::min<int64_t>((n + block - 1) / block, cap)); }

// split-K plan + launch (ws = workspace of ws_floats floats for the partial sums)
template <int A_K1, int B_N1>
void launch_gemm_f32(GemmF32 g, float* ws, size_t ws_floats, hipStream_t s) {
    // 128 x 128 tiles when K is long and the output still yields >= 64 of them (conv2 / conv3 forward, dgrad, wgrad), else 64 x 64
    const bool big = g.K >= 256 && ((g.M + 127) / 128) * ((g.N + 127) / 128) >= 64;
    const int TM = big ? 128 : 64;
    const int tiles = ((g.M + TM - 1) / TM) * ((g.N + TM - 1) / TM);
    const int ksteps = (g.K + TBK - 1) / TBK;
    int splits = std::max(1, std::min((big ? 768 : 1536) / std::max(tiles, 1), ksteps / 8));
    while (splits > 1 && (size_t)splits * g.M * g.N > ws_floats) --splits;
    int k_per = ((ksteps + splits - 1) / splits) * TBK;
    splits = (g.K + k_per - 1) / k_per;
    g.k_per_split = k_per;
    g.splits = splits;
    float* out = g.C;
    const int64_t ldc = g.ldc;
    const float* bias = g.bias;
    if (splits > 1) { g.C = ws; g.ldc = g.N; g.bias = nullptr; }
    const dim3 grid((g.N + TM - 1) / TM, (g.M + TM - 1) / TM, splits);
    if (big) hipLaunchKernelGGL((k_gemm_f32<A_K1, B_N1, 128>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_gemm_f32<A_K1, B_N1, 64>), grid, dim3(256), 0, s, g);
    if (splits > 1)
        launch_splitk_reduce(ws, splits, g.M, g.N, out, ldc, bias, s);
}

// C[M][N] = A[M][K] W[K][N] + bias
void gemm_nn(const float* A, int64_t lda, const float* W, float* Cm, const float* bias, int M, int N, int K, float* ws, size_t wsn,
             hipStream_t s, bool dma = false) {
    GemmF32 g{A, lda, 1, W, N, 1, Cm, N, bias, M, N, K, 0, 1};
    if (dma && M >= 128 && K % GD_BK == 0 && K >= 8 * GD_BK && N % 128 == 0 && lda % 4 == 0) {
        // k_gemm_f32_dma: 128 x 128 tiles, two workgroups per CU (512 slots a round), at least 8 K-steps per slice
        const int tiles = ((M + 127) / 128) * (N / 128), ksteps = K / GD_BK;
        int splits = std::max(1, std::min(512 / tiles, ksteps / 8));      // never a second round of a few workgroups
        while (splits > 1 && (size_t)splits * M * N > wsn) --splits;
        const int k_per = ((ksteps + splits - 1) / splits) * GD_BK;
        splits = (K + k_per - 1) / k_per;
        g.k_per_split = k_per;
        g.splits = splits;
        if (splits > 1) { g.C = ws; g.bias = nullptr; }
        hipLaunchKernelGGL(k_gemm_f32_dma, dim3((unsigned)((tiles * splits + 7) / 8 * 8)), dim3(256), 0, s, g);
        if (splits > 1) launch_splitk_reduce(ws, splits, M, N, Cm, (int64_t)N, bias, s);
        return;
    }
    launch_gemm_f32<1, 1>(g, ws, wsn, s);
}
// dA[M][K] = dZ[M][N] W[K][N]^T   (contraction over n; "B"(n, k) = W[k*N + n])
void gemm_nt(const float* dZ, const float* W, float* dA, int64_t ldda, int M, int N, int K, float* ws, size_t wsn, hipStream_t s) {
    GemmF32 g{dZ, N, 1, W, 1, N, dA, ldda, nullptr, M, K, N, 0, 1};
    launch_gemm_f32<1, 0>(g, ws, wsn, s);
}
// dW[K][N] = A[M][K]^T dZ[M][N]   (contraction over rows; "A"(k, r) = A[r*lda + k])
void gemm_tn(const float* A, int64_t lda, const float* dZ, float* dW, int M, int N, int K, float* ws, size_t wsn, hipStream_t s) {
    GemmF32 g{A, 1, lda, dZ, N, 1, dW, N, nullptr, K, N, M, 0, 1};
    launch_gemm_f32<0, 1>(g, ws, wsn, s);
}

int red_parts(int M) { return std::max(1, std::min(RED_PARTS, (M + 63) / 64)); }

// out[M][N] (row stride ldo) = A W^T (+ bias) as bf16 x 3: A hi / lo [M][lda], W hi / lo [N][ldw], both contiguous along the contraction
// Kc.  Split-K so that about two workgroups per CU exist (a workgroup walks at least 8 K-steps); the slices are summed in slice order.
void launch_gemm3(const uint16_t* a_hi, const uint16_t* a_lo, int lda, const uint16_t* w_hi, const uint16_t* w_lo, int ldw, float* out, int ldo,
                  const float* bias, int M, int N, int Kc, float* ws, size_t ws_floats, hipStream_t s, bool ring = false, bool f16 = false,
                  float out_scale = 1.0f, const ImplicitA* ia = nullptr) {
    const int steps = Kc / 32;
    const int mt_r = (M + G3R_BM - 1) / G3R_BM;
    // at most a tenth of the row tiles' rows beyond M, and a contraction long enough to fill and drain the ring (conv4's wgrad has 12
    // K-steps: 14.9 us on k_gemm3, 15.8 on the ring)
    if (ia || (ring && steps >= 16 && (mt_r * G3R_BM - M) * 10 <= mt_r * G3R_BM)) {      // a gathered A exists on the ring kernel only
        // k_gemm3_ring: 256 x 128 tiles, ONE workgroup per CU: never more than 256 of them while the tiles fit (a second round of a few
        // workgroups costs a whole round), at least 6 K-steps per slice (the ring is 3 deep)
        const int mt = mt_r, NT = N / 128, tiles = mt * NT;
        int splits = std::max(1, std::min(256 / tiles, steps / 6));
        while (splits > 1 && (size_t)splits * M * N > ws_floats) --splits;
        const int sps = (steps + splits - 1) / splits;
        splits = (steps + sps - 1) / sps;
        Gemm3 g{a_hi, a_lo, w_hi, w_lo, splits > 1 ? ws : out, bias, M, N, Kc, lda, ldw, ldo, sps, splits, 0, out_scale, ia ? *ia : ImplicitA{}};
        if (f16) hipLaunchKernelGGL((k_gemm3_ring<true>), dim3((unsigned)((tiles * splits + 7) / 8 * 8)), dim3(512), 0, s, g);
        else hipLaunchKernelGGL((k_gemm3_ring<false>), dim3((unsigned)((tiles * splits + 7) / 8 * 8)), dim3(512), 0, s, g);
        if (splits > 1)
            launch_splitk_reduce(ws, splits, M, N, out, (int64_t)ldo, bias, s);
        return;
    }
    const int mt = (M + G3_BM - 1) / G3_BM, NT = N / 128, tiles = mt * NT;
    int splits = std::max(1, std::min((512 + tiles / 2) / tiles, steps / 8));
    while (splits > 1 && (size_t)splits * M * N > ws_floats) --splits;
    const int sps = (steps + splits - 1) / splits;
    splits = (steps + sps - 1) / sps;
    const int xcd_rows = mt >= 8 ? 1 : 0;
    Gemm3 g{a_hi, a_lo, w_hi, w_lo, splits > 1 ? ws : out, bias, M, N, Kc, lda, ldw, ldo, sps, splits, xcd_rows, out_scale, ImplicitA{}};
    const dim3 grid((unsigned)((xcd_rows ? (mt + 7) / 8 * 8 : mt) * NT), (unsigned)splits);
    if (f16) hipLaunchKernelGGL((k_gemm3<true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_gemm3<false>), grid, dim3(256), 0, s, g);
    if (splits > 1)
        launch_splitk_reduce(ws, splits, M, N, out, (int64_t)ldo, bias, s);
}
// dW [Kin][N] = A^T dz on k_wgrad3_tr; split-K over the 32-row steps so that at most 256 workgroups exist, slices summed in slice order
void launch_wgrad3_tr(const uint16_t* a_hi, const uint16_t* a_lo, int lda, const uint16_t* z_hi, const uint16_t* z_lo, int ldz, float* out, int M,
                      int Kin, int N, float* ws, size_t ws_floats, hipStream_t s, const ImplicitA* ia = nullptr) {
    const int tiles = (Kin / 256) * (N / 128), steps = M / 32;
    int splits = std::max(1, std::min(256 / tiles, steps / 6));
    while (splits > 1 && (size_t)splits * Kin * N > ws_floats) --splits;
    const int sps = (steps + splits - 1) / splits;
    splits = (steps + sps - 1) / sps;
    Wgrad3 g{a_hi, a_lo, z_hi, z_lo, splits > 1 ? ws : out, Kin, N, M, lda, ldz, sps, splits, ia ? *ia : ImplicitA{},
             ia ? (65536 + ia->Ht * ia->Wt - 1) / (ia->Ht * ia->Wt) : 0, ia ? (65536 + ia->Wt - 1) / ia->Wt : 0};
    hipLaunchKernelGGL(k_wgrad3_tr, dim3((unsigned)((tiles * splits + 7) / 8 * 8)), dim3(512), 0, s, g);
    if (splits > 1) launch_splitk_reduce(ws, splits, Kin, N, out, (int64_t)N, nullptr, s);
}
void launch_transpose_split2(const TransposeJob& j0, const TransposeJob& j1, int R, int Rp, hipStream_t s) {
    hipLaunchKernelGGL((k_transpose_split<false>), dim3((unsigned)(j0.tiles_c + j1.tiles_c), (unsigned)(Rp / 64)), dim3(256), 0, s, j0, j1,
                       TransposeJob{}, R, Rp, 1.0f);
}

}  // namespace

namespace {

// every kernel of one optimisation step, in order, on stream s (no host synchronisation: capturable in a hipGraph)
void enqueue_step(Trainer* t, const TrainHyper& h, const float* d_boards, const float* d_pis, const float* d_vs, int b, bool apply,
                  hipStream_t s, bool col1_done = false) {
    const int C = t->C;
    const Layout& L = t->L;
    float* P = t->params;
    float* G = t->grads;
    const StepState* st = t->step_state;
    struct LayerDef { const float* A; int64_t lda; int M, K, N; int64_t w, bias, bn; bool drop; };
    const int rows[6] = {b * 42, b * 42, b * 20, b * 6, b, b};
    const LayerDef ld[6] = {
        {t->col[0], 20, rows[0], 18, C, L.conv_w[0], L.conv_b[0], L.conv_bn[0], false},
        {t->col[1], 9ll * C, rows[1], 9 * C, C, L.conv_w[1], L.conv_b[1], L.conv_bn[1], false},
        {t->col[2], 9ll * C, rows[2], 9 * C, C, L.conv_w[2], L.conv_b[2], L.conv_bn[2], false},
        {t->col[3], 9ll * C, rows[3], 9 * C, C, L.conv_w[3], L.conv_b[3], L.conv_bn[3], false},
        {t->a[3], 6ll * C, b, 6 * C, 1024, L.fc_w[0], L.fc_b[0], L.fc_bn[0], true},
        {t->a[4], 1024, b, 1024, 512, L.fc_w[1], L.fc_b[1], L.fc_bn[1], true},
    };
    const float keep = 1.0f - h.dropout;
    const uint32_t keep_thresh = h.dropout > 0.0f ? (uint32_t)(keep * 16777216.0f) : 0u;
    const float drop_scale = h.dropout > 0.0f ? 1.0f / keep : 1.0f;
    auto bn_desc = [&](int l, float* out, const float* grad_out) {
        BnLayer d{};
        d.z = t->z[l]; d.out = out; d.grad_out = grad_out;
        d.gamma = P + ld[l].bn; d.beta = P + ld[l].bn + ld[l].N;
        d.mean = t->mean[l]; d.invstd = t->invstd[l];
        d.M = ld[l].M; d.N = ld[l].N;
        d.drop_layer = (uint32_t)l;
        d.keep_thresh = ld[l].drop ? keep_thresh : 0u;
        d.drop_scale = drop_scale;
        return d;
    };
    constexpr int APPLY_ROWS = 32;      // rows per block of the apply kernels
    // ---- forward ----
    const bool x3 = t->gemm_mode == 1;
    const bool fork = x3 && t->fork && t->side;
    // the gathered conv GEMMs need the f16 x 3 forward, k_wgrad3_tr's shapes (42 b, 20 b, 6 b rows in 32s: b % 16 == 0; C % 256 == 0)
    const bool impl = x3 && t->implicit && t->fwd_x3 && t->wgrad_tr && b % 16 == 0 && b * 20 > BN_SMALL_ROWS && C % 256 == 0;
    // the FC layers' wgrad on k_wgrad3_tr as well (A = the previous layer's activations as bf16 pairs, two 32-row K-steps at batch 64)
    auto fc_tr = [&](int l) { return x3 && t->wgrad_tr && l >= 4 && b % 32 == 0 && ld[l].K % 256 == 0 && ld[l].N % 128 == 0; };
    const int geo[4][3] = {{6, 7, 1}, {6, 7, 1}, {6, 7, 0}, {4, 5, 0}};      // conv layer l: input H, W, pad
    auto gather = [&](int l, bool dgrad) {      // ImplicitA of conv layer l (1..3): forward / wgrad rows = outputs, dgrad rows = inputs
        const int H = geo[l][0], W = geo[l][1], pad = geo[l][2], Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
        ImplicitA ia{};
        ia.on = 1; ia.Cs = C;
        if (!dgrad) { ia.Ht = Ho; ia.Wt = Wo; ia.Hs = H; ia.Ws = W; ia.sgn = 1; ia.off = -pad; ia.zero_off = t->act_zero_off[l - 1]; }
        else { ia.Ht = H; ia.Wt = W; ia.Hs = Ho; ia.Ws = Wo; ia.sgn = -1; ia.off = pad; ia.zero_off = t->dz_zero_off; }
        return ia;
    };
    hipStream_t s2 = fork ? t->side : s;
    auto hand = [&](hipEvent_t ev, hipStream_t from, hipStream_t to) {       // `to` continues after everything enqueued on `from` so far
        if (fork) { (void)hipEventRecord(ev, from); (void)hipStreamWaitEvent(to, ev, 0); }
    };
    auto split_desc = [&]() {
        SplitWeights sw{};
        const int64_t offs[5] = {L.conv_w[1], L.conv_w[2], L.conv_w[3], L.fc_w[0], L.fc_w[1]};
        for (int i = 0; i < 5; ++i) { sw.off[i] = offs[i]; sw.count[i] = (int64_t)ld[i + 1].K * ld[i + 1].N; }
        if (impl) { sw.perm_c[0] = C; sw.perm_n[0] = C; }      // conv2 only: its dgrad is the gathered GEMM
        return sw;
    };
    auto split_weights = [&]() {
        const SplitWeights sw = split_desc();
        hipLaunchKernelGGL(k_split_weights, dim3(256, 5), dim3(256), 0, s2, (const float*)P, sw, t->w_hi, t->w_lo);
    };
    if (fork) {                      // the weight split needs the parameters only: under the forward pass
        hand(t->ev_start, s, s2);
        split_weights();
        (void)hipEventRecord(t->ev_sw, s2);
    }
    if (!col1_done) hipLaunchKernelGGL(k_boards_col1, grid1((int64_t)b * 42 * 20, 256, 1 << 20), dim3(256), 0, s, d_boards, t->col[0], b);
    const bool fx3 = x3 && t->fwd_x3;           // "train_gemm" 0 keeps every GEMM on the f32 matrix cores
    // k_wgrad3_tr's shape constraints (a batch of 64 meets them at every width that is a multiple of 256 / 9 ... i.e. 9 C % 256 == 0)
    auto tr_ok = [&](int l) { return x3 && t->wgrad_tr && l >= 1 && l <= 3 && ld[l].M % 32 == 0 && ld[l].K % 256 == 0 && ld[l].N % 128 == 0; };
    const bool prep_one = fx3 && !fork;      // both weight preparations in one launch, ahead of the forward pass
    if (fx3) {          // W [9C][C] of conv2..conv4 -> (256 W)^T as half hi / lo [C][9C]
        TransposeJob j[3];
        for (int l = 1; l <= 3; ++l) j[l - 1] = TransposeJob{P + ld[l].w, t->wt_hi[l], t->wt_lo[l], C, C / 64, C};
        const int tgx = 3 * (C / 64), tgy = 9 * C / 64;
        if (prep_one) {
            const SplitWeights sw = split_desc();
            hipLaunchKernelGGL(k_weight_prep, dim3((unsigned)(tgx * tgy + 256 * 5)), dim3(256), 0, s, j[0], j[1], j[2], 9 * C, 9 * C, 256.0f, tgx, tgy,
                               (const float*)P, sw, t->w_hi, t->w_lo, 256);
        } else {
            hipLaunchKernelGGL((k_transpose_split<true>), dim3((unsigned)tgx, (unsigned)tgy), dim3(256), 0, s, j[0], j[1], j[2], 9 * C, 9 * C, 256.0f);
        }
    }
    for (int l = 0; l < 6; ++l) {
        const LayerDef& d = ld[l];
        if (impl && l >= 1 && l <= 3) {
            const ImplicitA ia = gather(l, false);
            launch_gemm3(t->act_hi[l - 1], t->act_lo[l - 1], 0, t->wt_hi[l], t->wt_lo[l], 9 * C, t->z[l], d.N, P + d.bias, d.M, d.N, d.K, t->splitk,
                         t->splitk_floats, s, true, true, 1.0f / (256.0f * 64.0f), &ia);
        } else if (fx3 && l >= 1 && l <= 3)
            launch_gemm3(t->col_hi[l], t->col_lo[l], 9 * C, t->wt_hi[l], t->wt_lo[l], 9 * C, t->z[l], d.N, P + d.bias, d.M, d.N, d.K, t->splitk,
                         t->splitk_floats, s, t->gemm3_ring, true, 1.0f / (256.0f * 64.0f));
        else
            gemm_nn(d.A, d.lda, P + d.w, t->z[l], P + d.bias, d.M, d.N, d.K, t->splitk, t->splitk_floats, s, t->fwd_dma);
        BnLayer bn = bn_desc(l, t->a[l], nullptr);
        if (impl && l <= 2) { bn.act_hi = t->act_hi[l]; bn.act_lo = t->act_lo[l]; bn.act_bhi = t->act_bhi[l]; bn.act_blo = t->act_blo[l]; }
        if ((l == 3 || l == 4) && fc_tr(l + 1)) { bn.act_bhi = t->act_bhi[l]; bn.act_blo = t->act_blo[l]; }
        const int parts = red_parts(d.M), rpb = (d.M + parts - 1) / parts;
        if (d.M <= BN_SMALL_ROWS) {
            hipLaunchKernelGGL(k_bn_fwd_small, dim3(d.N / BN_COLS), dim3(256), 0, s, bn, h.bn_eps, h.bn_momentum, P + d.bn + 2 * d.N, P + d.bn + 3 * d.N, st);
        } else {
            hipLaunchKernelGGL((k_colreduce<0>), dim3(d.N / BN_COLS, parts), dim3(256), 0, s, bn, rpb, t->partial, st);
            hipLaunchKernelGGL(k_bn_apply, dim3(d.N / BN_COLS, (d.M + APPLY_ROWS - 1) / APPLY_ROWS), dim3(256), 0, s, bn, t->partial, parts,
                               APPLY_ROWS, h.bn_eps, h.bn_momentum, P + d.bn + 2 * d.N, P + d.bn + 3 * d.N, st);
        }
        if (l <= 2 && !impl) {
            const int ln = l + 1;                // the layer this matrix feeds
            Im2colOut io{};
            if (fx3) { io.col_hi = t->col_hi[ln]; io.col_lo = t->col_lo[ln]; }
            if (tr_ok(ln)) { io.col_bhi = t->col_bhi[ln]; io.col_blo = t->col_blo[ln]; }
            if (!fx3 || !tr_ok(ln)) io.col = t->col[ln];
            const int H = l == 2 ? 4 : 6, W = l == 2 ? 5 : 7, pad = l == 0 ? 1 : 0;
            hipLaunchKernelGGL(k_im2col, grid1((int64_t)ld[ln].M * 9 * C / 4), dim3(256), 0, s, t->a[l], io, b, H, W, C, pad);
        }
    }
    hipLaunchKernelGGL(k_heads_loss, dim3(b), dim3(64), 0, s, t->a[5], P + L.pi_w, P + L.pi_b, P + L.v_w, P + L.v_b, d_pis, d_vs, b,
                       t->dhead, t->sample_loss, t->logits);
    // ---- backward ----
    hipLaunchKernelGGL(k_heads_bwd, grid1(std::max<int64_t>(512 * 8 + 8, (int64_t)b * 512)), dim3(256), 0, s, t->a[5], t->dhead,
                       P + L.pi_w, P + L.v_w, b, G + L.pi_w, G + L.pi_b, G + L.v_w, G + L.v_b, t->dact, t->sample_loss, t->loss_totals);
    if (x3 && !fork && !prep_one) split_weights();
    if (fork) (void)hipStreamWaitEvent(s, t->ev_sw, 0);
    for (int l = 5; l >= 0; --l) {
        const LayerDef& d = ld[l];
        // t->dact holds d loss / d a[l]  ->  dz (through dropout, ReLU and BatchNorm)
        BnLayer bn = bn_desc(l, t->dz, t->dact);
        const bool g3 = x3 && l >= 1;
        if (g3) { bn.out_hi = t->dz_hi; bn.out_lo = t->dz_lo; }
        const int parts = red_parts(d.M), rpb = (d.M + parts - 1) / parts;
        if (fork && l <= 4) (void)hipStreamWaitEvent(s, t->ev_tr[l + 1], 0);      // dz is rewritten: the layer above has transposed it
        if (d.M <= BN_SMALL_ROWS) {
            hipLaunchKernelGGL(k_bn_bwd_small, dim3(d.N / BN_COLS), dim3(256), 0, s, bn, G + d.bn, G + d.bn + d.N, st);
        } else {
            hipLaunchKernelGGL((k_colreduce<1>), dim3(d.N / BN_COLS, parts), dim3(256), 0, s, bn, rpb, t->partial, st);
            hipLaunchKernelGGL(k_bn_bwd_apply, dim3(d.N / BN_COLS, (d.M + APPLY_ROWS - 1) / APPLY_ROWS), dim3(256), 0, s, bn, t->partial,
                               parts, APPLY_ROWS, G + d.bn, G + d.bn + d.N, st);
        }
        // The gradient of a bias in front of a BatchNorm is identically zero (the batch mean absorbs it); the kernels
        // leave those slots at 0 instead of the rounding residue a column sum of dz would give, which Adam would
        // turn into a random walk of size lr.
        if (g3) {
            // wgrad: dW [K][N] = A^T dz, both operands transposed so that the contraction (the rows) is contiguous
            const int Mp = (d.M + 63) / 64 * 64;
            hand(t->ev_dz[l], s, s2);
            if (impl && l <= 3) {     // A gathered from the layer's input, dz as stored
                const ImplicitA ia = gather(l, false);
                launch_wgrad3_tr(t->act_bhi[l - 1], t->act_blo[l - 1], 0, t->dz_hi, t->dz_lo, d.N, G + d.w, d.M, d.K, d.N, fork ? t->splitk2 : t->splitk,
                                 fork ? t->splitk2_floats : t->splitk_floats, s2, &ia);
                if (fork) (void)hipEventRecord(t->ev_tr[l], s2);
            } else if (fc_tr(l)) {
                launch_wgrad3_tr(t->act_bhi[l - 1], t->act_blo[l - 1], d.K, t->dz_hi, t->dz_lo, d.N, G + d.w, d.M, d.K, d.N, fork ? t->splitk2 : t->splitk,
                                 fork ? t->splitk2_floats : t->splitk_floats, s2);
                if (fork) (void)hipEventRecord(t->ev_tr[l], s2);
            } else if (tr_ok(l)) {       // A and dz as stored: no transposes
                launch_wgrad3_tr(t->col_bhi[l], t->col_blo[l], (int)d.lda, t->dz_hi, t->dz_lo, d.N, G + d.w, d.M, d.K, d.N, fork ? t->splitk2 : t->splitk,
                                 fork ? t->splitk2_floats : t->splitk_floats, s2);
                if (fork) (void)hipEventRecord(t->ev_tr[l], s2);
            } else {
                const TransposeJob jz{t->dz, t->dzt_hi, t->dzt_lo, d.N, (d.N + 63) / 64, d.N}, ja{d.A, t->at_hi, t->at_lo, d.K, (d.K + 63) / 64, d.lda};
                launch_transpose_split2(ja, jz, d.M, Mp, s2);         // the big one first
                if (fork) (void)hipEventRecord(t->ev_tr[l], s2);
                launch_gemm3(t->at_hi, t->at_lo, Mp, t->dzt_hi, t->dzt_lo, Mp, G + d.w, d.N, nullptr, d.K, d.N, Mp, fork ? t->splitk2 : t->splitk,
                             fork ? t->splitk2_floats : t->splitk_floats, s2, t->gemm3_ring);
            }
            // dgrad: d input [M][K] = dz W^T, W [K][N] as stored
            float* din = l >= 4 ? t->dact : t->dcol;
            // d input [b Hin Win][C] = the transposed convolution as ONE GEMM over (tap, n): no dcol, no col2im -- for the 'same' convolution
            // (conv2) only: a 'valid' one has fewer output than input positions, and the gathered form multiplies every input position by
            // all nine taps (conv3 2.1 x, conv4 3.3 x the products of dz W^T + col2im; measured 49 vs 41 and 34 vs 24 us)
            if (impl && l == 1) {
                const ImplicitA ia = gather(l, true);
                ImplicitA ig = ia;
                ig.Cs = d.N;
                launch_gemm3(t->dz_hi, t->dz_lo, 0, t->w_hi + d.w, t->w_lo + d.w, 9 * d.N, t->dact, C, nullptr, b * ia.Ht * ia.Wt, C, 9 * d.N, t->splitk,
                             t->splitk_floats, s, true, false, 1.0f, &ig);
            } else
            launch_gemm3(t->dz_hi, t->dz_lo, d.N, t->w_hi + d.w, t->w_lo + d.w, d.N, din, d.K, nullptr, d.M, d.K, d.N, t->splitk, t->splitk_floats, s,
                         t->gemm3_ring);
        } else {
            gemm_tn(d.A, d.lda, t->dz, G + d.w, d.M, d.N, d.K, t->splitk, t->splitk_floats, s);
            if (l == 0) break;
            if (l >= 4) gemm_nt(t->dz, P + d.w, t->dact, d.K, d.M, d.N, d.K, t->splitk, t->splitk_floats, s);      // FC: d a[l-1] directly
            else gemm_nt(t->dz, P + d.w, t->dcol, d.K, d.M, d.N, d.K, t->splitk, t->splitk_floats, s);
        }
        if (l >= 1 && l <= 3 && !(impl && g3 && l == 1)) {
            const int H = l == 3 ? 4 : 6, W = l == 3 ? 5 : 7, pad = l == 1 ? 1 : 0;
            hipLaunchKernelGGL(k_col2im, grid1((int64_t)b * H * W * C / 4), dim3(256), 0, s, t->dcol, t->dact, b, H, W, C, pad);
        }
    }
    hand(t->ev_join, s2, s);
    if (apply)
        hipLaunchKernelGGL(k_adam, grid1(L.total), dim3(256), 0, s, P, G, t->m, t->v, L.total, h.lr, h.beta1, h.beta2, h.adam_eps, st);
}

}  // namespace

bool trainer_step(Trainer* t, const TrainHyper& h, const float* d_boards, const float* d_pis, const float* d_vs, int b,
                  uint64_t mask_seed, bool apply, hipStream_t s) {
    if (!t || b <= 1 || b > TRAIN_MAX_BATCH) return false;
    StepState& st = t->host_state[t->host_state_next++ % HOST_STATE_RING];
    st.mask_seed = mask_seed;
    const double p1 = t->pow1 * (double)h.beta1, p2 = t->pow2 * (double)h.beta2;
    st.bc1 = (float)(1.0 - p1);
    st.sqrt_bc2 = sqrtf((float)(1.0 - p2));
    st.idx_offset = 0;
    if (hipMemcpyAsync(t->step_state, &st, sizeof st, hipMemcpyHostToDevice, s) != hipSuccess) return false;
    enqueue_step(t, h, d_boards, d_pis, d_vs, b, apply, s);
    if (apply) { t->step += 1; t->pow1 = p1; t->pow2 = p2; }
    return hipGetLastError() == hipSuccess;
}

bool trainer_run_epoch(Trainer* t, const TrainHyper& h, const float* all_boards, const float* all_pis, const float* all_vs,
                       const int64_t* d_idx, int64_t steps, int b, uint64_t seed_key, uint64_t gstep0, hipStream_t s) {
    if (!t || b <= 1 || b > TRAIN_MAX_BATCH || steps <= 0) return false;
    EpochCounters c{seed_key, gstep0, 0, (double)h.beta1, (double)h.beta2, t->pow1, t->pow2};
    if (hipMemcpyAsync(t->counters, &c, sizeof c, hipMemcpyHostToDevice, s) != hipSuccess) return false;
    if (hipStreamSynchronize(s) != hipSuccess) return false;      // `c` is a stack object
    auto one_step = [&]() {
        hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(64), 0, s, t->step_state, t->counters, b);
        hipLaunchKernelGGL(k_gather_col1, dim3((b * 42 * 20 + 255) / 256), dim3(256), 0, s, all_boards, all_pis, all_vs, d_idx, b, t->bboards,
                           t->bpis, t->bvs, t->col[0], (const StepState*)t->step_state);
        enqueue_step(t, h, t->bboards, t->bpis, t->bvs, b, true, s, true);
    };
    // the step's ~60 launches are captured once and replayed: the epoch is launch-bound otherwise
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool graphed = false;
    if (t->use_graph && hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        one_step();
        if (hipStreamEndCapture(s, &graph) == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess)
            graphed = true;
    }
    (void)hipGetLastError();
    bool ok = true;
    for (int64_t i = 0; i < steps && ok; ++i) {
        if (graphed) ok = hipGraphLaunch(exec, s) == hipSuccess;
        else one_step();
    }
    ok = ok && hipStreamSynchronize(s) == hipSuccess && hipGetLastError() == hipSuccess;
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    t->step += steps;
    for (int64_t i = 0; i < steps; ++i) { t->pow1 *= (double)h.beta1; t->pow2 *= (double)h.beta2; }
    return ok;
}

void trainer_set_graph(Trainer* t, bool on) { if (t) t->use_graph = on; }
void trainer_set_gemm(Trainer* t, int mode) { if (t) t->gemm_mode = mode; }
// The branch's stream and events exist only once the option has been switched on: an idle extra stream still takes one of the device's
// few hardware queues, and the arena's two search streams then share one (measured: az_arena 2755 -> 1860 games/s in a process whose
// trainer merely owned a second stream).
void trainer_set_fork(Trainer* t, bool on) {
    if (!t) return;
    if (on && !t->side) {
        bool ok = hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking) == hipSuccess;
        for (hipEvent_t* ev : {&t->ev_start, &t->ev_sw, &t->ev_join}) ok = ok && hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess;
        for (int l = 0; l < 6 && ok; ++l)
            ok = hipEventCreateWithFlags(&t->ev_dz[l], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&t->ev_tr[l], hipEventDisableTiming) == hipSuccess;
        if (!ok) { if (t->side) { (void)hipStreamDestroy(t->side); t->side = nullptr; } on = false; }
    }
    t->fork = on;
}
void trainer_set_fwd_x3(Trainer* t, bool on) { if (t) t->fwd_x3 = on; }
void trainer_set_wgrad_tr(Trainer* t, bool on) { if (t) t->wgrad_tr = on; }
void trainer_set_implicit(Trainer* t, bool on) { if (t) t->implicit = on; }
void trainer_set_gemm3_ring(Trainer* t, bool on) { if (t) t->gemm3_ring = on; }
void trainer_set_fwd_dma(Trainer* t, bool on) { if (t) t->fwd_dma = on; }

}  // namespace az
