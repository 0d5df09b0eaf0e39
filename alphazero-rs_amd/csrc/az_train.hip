// az_train.hip -- NNet::train (src/nnet.rs:38) as f32 HIP kernels for gfx950.
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
// Everything is f32 (activations row-major [rows][channels], rows = (sample, y, x)); the GEMMs use
// v_mfma_f32_16x16x4_f32 with 64x64 block tiles.  At the reference's batch of 64 a step is ~63 GFLOP over ~45
// small launches: it is launch- and fill-bound, not roofline-bound, and is sized for correctness first
// (DESIGN.md section 8).
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
// k-major in LDS (As[k][m], Bs[k][n], row stride 80 floats: the four k rows a wave reads sit 16 banks apart).
struct GemmF32 {
    const float* A; int64_t sAm, sAk;
    const float* B; int64_t sBk, sBn;
    float* C; int64_t ldc;
    const float* bias;      // nullptr = none
    int M, N, K;
};

constexpr int TBM = 64, TBN = 64, TBK = 16, TLD = 80;

AZ_D float4 ld4_guard(const float* p, int64_t stride, int valid) {   // up to 4 elements p[0], p[stride], ...
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid >= 4 && stride == 1 && (((uintptr_t)p) & 15) == 0) return *(const float4*)p;
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[stride];
    if (valid > 2) v.z = p[2 * stride];
    if (valid > 3) v.w = p[3 * stride];
    return v;
}

template <int A_K1, int B_N1>
__global__ __launch_bounds__(256) void k_gemm_f32(const GemmF32 g) {
    __shared__ __attribute__((aligned(16))) float As[TBK][TLD];
    __shared__ __attribute__((aligned(16))) float Bs[TBK][TLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * TBM, n0 = blockIdx.x * TBN;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;
    for (int k0 = 0; k0 < g.K; k0 += TBK) {
        if (A_K1) {        // 64 rows x 16 k: thread -> row tid>>2, k quad (tid&3)*4; stored transposed
            const int m = tid >> 2, kq = (tid & 3) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + m < g.M) v = ld4_guard(g.A + (int64_t)(m0 + m) * g.sAm + (k0 + kq), 1, g.K - (k0 + kq));
            As[kq + 0][m] = v.x; As[kq + 1][m] = v.y; As[kq + 2][m] = v.z; As[kq + 3][m] = v.w;
        } else {           // 16 k x 64 rows: thread -> k tid>>4, row quad (tid&15)*4
            const int k = tid >> 4, mq = (tid & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + k < g.K) v = ld4_guard(g.A + (int64_t)(k0 + k) * g.sAk + (m0 + mq), 1, g.M - (m0 + mq));
            *(float4*)&As[k][mq] = v;
        }
        if (B_N1) {
            const int k = tid >> 4, nq = (tid & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + k < g.K) v = ld4_guard(g.B + (int64_t)(k0 + k) * g.sBk + (n0 + nq), 1, g.N - (n0 + nq));
            *(float4*)&Bs[k][nq] = v;
        } else {
            const int n = tid >> 2, kq = (tid & 3) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + n < g.N) v = ld4_guard(g.B + (int64_t)(n0 + n) * g.sBn + (k0 + kq), 1, g.K - (k0 + kq));
            Bs[kq + 0][n] = v.x; Bs[kq + 1][n] = v.y; Bs[kq + 2][n] = v.z; Bs[kq + 3][n] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = kk * 4 + fk;
            const float a0 = As[k][wm * 32 + fr], a1 = As[k][wm * 32 + 16 + fr];
            const float b0 = Bs[k][wn * 32 + fr], b1 = Bs[k][wn * 32 + 16 + fr];
            // the B tile is the instruction's first operand: D[n][m], a lane holds 4 consecutive n of one row m
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0, a0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, a0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0, a1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, a1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 32 + i * 16 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 32 + j * 16 + fk * 4;
            float* c = g.C + (int64_t)m * g.ldc + n;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (n + q < g.N) c[q] = acc[i][j][q] + (g.bias ? g.bias[n + q] : 0.0f);
        }
    }
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
__global__ void k_im2col(const float* __restrict__ in, float* __restrict__ col, int b, int H, int W, int C, int pad) {
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
        *(float4*)(col + (size_t)row * 9 * C + (size_t)tap * C + c4 * 4) = v;
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

__global__ void k_gather_batch(const float* __restrict__ all_boards, const float* __restrict__ all_pis,
                               const float* __restrict__ all_vs, const int64_t* __restrict__ idx, int b,
                               float* __restrict__ boards, float* __restrict__ pis, float* __restrict__ vs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * 92) return;
    const int j = i / 92, f = i % 92;
    const int64_t src = idx[j];
    if (f < 84) boards[(size_t)j * 84 + f] = all_boards[(size_t)src * 84 + f];
    else if (f < 91) pis[(size_t)j * 7 + (f - 84)] = all_pis[(size_t)src * 7 + (f - 84)];
    else vs[j] = all_vs[src];
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
    uint64_t mask_seed;
};

AZ_D float bn_grad_in(const BnLayer& L, int r, int c, float mean, float invstd, float gamma, float beta, float& xh) {
    const size_t i = (size_t)r * L.N + c;
    xh = (L.z[i] - mean) * invstd;
    const float y = gamma * xh + beta;
    float g = y > 0.0f ? L.grad_out[i] : 0.0f;
    if (L.keep_thresh) g = dropout_keep(L.mask_seed, L.drop_layer, i, L.keep_thresh) ? g * L.drop_scale : 0.0f;
    return g;
}

// stage 1 of a column reduction: block (64 columns x 4 row lanes) reduces rows [blockIdx.y*rpb, +rpb) and writes
// partial[blockIdx.y][column][2] (f64); KIND 0: (sum z, sum z^2); 1: BN backward (sum g, sum g*xhat); 2: (sum x, -)
template <int KIND>
__global__ __launch_bounds__(256) void k_colreduce(const BnLayer L, const float* __restrict__ x, int rpb, double* __restrict__ partial) {
    __shared__ double red[4][64][2];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rpb, r1 = min(L.M, r0 + rpb);
    double s0 = 0.0, s1 = 0.0;
    if (c < L.N) {
        float mean = 0.f, invstd = 0.f, gamma = 0.f, beta = 0.f;
        if (KIND == 1) { mean = L.mean[c]; invstd = L.invstd[c]; gamma = L.gamma[c]; beta = L.beta[c]; }
        for (int r = r0 + ty; r < r1; r += 4) {
            if (KIND == 0) { const float v = x[(size_t)r * L.N + c]; s0 += v; s1 += (double)v * v; }
            else if (KIND == 1) { float xh; const float g = bn_grad_in(L, r, c, mean, invstd, gamma, beta, xh); s0 += g; s1 += (double)g * xh; }
            else s0 += x[(size_t)r * L.N + c];
        }
    }
    red[ty][threadIdx.x & 63][0] = s0;
    red[ty][threadIdx.x & 63][1] = s1;
    __syncthreads();
    if (ty == 0 && c < L.N) {
        const int t = threadIdx.x;
        s0 = red[0][t][0] + red[1][t][0] + red[2][t][0] + red[3][t][0];
        s1 = red[0][t][1] + red[1][t][1] + red[2][t][1] + red[3][t][1];
        partial[((size_t)blockIdx.y * L.N + c) * 2 + 0] = s0;
        partial[((size_t)blockIdx.y * L.N + c) * 2 + 1] = s1;
    }
}

// stage 2, forward statistics: batch mean / biased variance -> mean, invstd; moving averages updated in place
// (moving = momentum*moving + (1-momentum)*batch, the variance with Bessel's correction, as F.batch_norm does)
__global__ void k_bn_stats_finish(const double* __restrict__ partial, int nparts, int M, int N, float eps, float momentum,
                                  float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ run_mean,
                                  float* __restrict__ run_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    double s0 = 0.0, s1 = 0.0;
    for (int p = 0; p < nparts; ++p) { s0 += partial[((size_t)p * N + c) * 2]; s1 += partial[((size_t)p * N + c) * 2 + 1]; }
    const double mu = s0 / M;
    double var = s1 / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    const double unbiased = M > 1 ? var * M / (M - 1) : var;
    run_mean[c] = momentum * run_mean[c] + (1.0f - momentum) * (float)mu;
    run_var[c] = momentum * run_var[c] + (1.0f - momentum) * (float)unbiased;
}

// stage 2, backward: dgamma = sum g*xhat, dbeta = sum g (also kept in sums[N][2] for k_bn_bwd_apply)
__global__ void k_bn_bwd_finish(const double* __restrict__ partial, int nparts, int N, float* __restrict__ sums,
                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    double s0 = 0.0, s1 = 0.0;
    for (int p = 0; p < nparts; ++p) { s0 += partial[((size_t)p * N + c) * 2]; s1 += partial[((size_t)p * N + c) * 2 + 1]; }
    sums[2 * c] = (float)s0; sums[2 * c + 1] = (float)s1;
    dbeta[c] = (float)s0; dgamma[c] = (float)s1;
}

__global__ void k_colsum_finish(const double* __restrict__ partial, int nparts, int N, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    double s0 = 0.0;
    for (int p = 0; p < nparts; ++p) s0 += partial[((size_t)p * N + c) * 2];
    out[c] = (float)s0;
}

// a = dropout(relu(gamma * xhat + beta))
__global__ void k_bn_apply(const BnLayer L) {
    const int64_t total = (int64_t)L.M * L.N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % L.N);
        const float xh = (L.z[i] - L.mean[c]) * L.invstd[c];
        float y = fmaxf(L.gamma[c] * xh + L.beta[c], 0.0f);
        if (L.keep_thresh) y = dropout_keep(L.mask_seed, L.drop_layer, (uint64_t)i, L.keep_thresh) ? y * L.drop_scale : 0.0f;
        L.out[i] = y;
    }
}

// dz = gamma * invstd * (g - (dbeta + xhat * dgamma) / M)
__global__ void k_bn_bwd_apply(const BnLayer L, const float* __restrict__ sums) {
    const int64_t total = (int64_t)L.M * L.N;
    const float inv_m = 1.0f / (float)L.M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % L.N), r = (int)(i / L.N);
        const float mean = L.mean[c], invstd = L.invstd[c], gamma = L.gamma[c];
        float xh;
        const float g = bn_grad_in(L, r, c, mean, invstd, gamma, L.beta[c], xh);
        L.out[i] = gamma * invstd * (g - (sums[2 * c] + xh * sums[2 * c + 1]) * inv_m);
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

// fixed-order sum of the per-sample losses into the running totals (one thread: b <= 256)
__global__ void k_loss_accumulate(const float* __restrict__ sample_loss, int b, double* __restrict__ totals) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double lp = 0.0, lv = 0.0;
    for (int j = 0; j < b; ++j) { lp += sample_loss[2 * j]; lv += sample_loss[2 * j + 1]; }
    totals[0] += lp / b;
    totals[1] += lv / b;
}

// d pi_w[k][o] = sum_j a[j][k] dhead[j][o], d v_w[k] = sum_j a[j][k] dhead[j][7]; biases = column sums of dhead;
// da[j][k] = sum_o dhead[j][o] pi_w[k][o] + dhead[j][7] v_w[k]
__global__ void k_heads_bwd(const float* __restrict__ a, const float* __restrict__ dhead, const float* __restrict__ pi_w,
                            const float* __restrict__ v_w, int b, float* __restrict__ d_pi_w, float* __restrict__ d_pi_b,
                            float* __restrict__ d_v_w, float* __restrict__ d_v_b, float* __restrict__ da) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 512 * 8) {
        const int k = i >> 3, o = i & 7;
        float s = 0.0f;
        for (int j = 0; j < b; ++j) s += a[(size_t)j * 512 + k] * dhead[(size_t)j * 8 + o];
        if (o < 7) d_pi_w[k * 7 + o] = s; else d_v_w[k] = s;
    } else if (i < 512 * 8 + 8) {
        const int o = i - 512 * 8;
        float s = 0.0f;
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
                       int64_t n, float lr, float b1, float b2, float eps, float bc1, float sqrt_bc2) {
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
    int64_t step = 0;
    template <class T> T* dalloc(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        dev.push_back(p);
        return (T*)p;
    }
};

constexpr int RED_PARTS = 32;

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
    if (!ok) { if (err) *err = "hipMalloc failed for the trainer workspace"; trainer_destroy(t); return nullptr; }
    (void)hipMemset(t->loss_totals, 0, 2 * sizeof(double));
    return t;
}

void trainer_destroy(Trainer* t) {
    if (!t) return;
    for (void* p : t->dev) (void)hipFree(p);
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

void trainer_gather(Trainer* t, const float* all_boards, const float* all_pis, const float* all_vs, const int64_t* d_idx, int b,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_gather_batch, dim3((b * 92 + 255) / 256), dim3(256), 0, s, all_boards, all_pis, all_vs, d_idx, b,
                       t->bboards, t->bpis, t->bvs);
}

namespace {

inline dim3 grid1(int64_t n, int block = 256, int cap = 4096) { return dim3((unsigned)std::min<int64_t>((n + block - 1) / block, cap)); }

// C[M][N] = A[M][K] W[K][N] + bias
void gemm_nn(const float* A, int64_t lda, const float* W, float* Cm, const float* bias, int M, int N, int K, hipStream_t s) {
    GemmF32 g{A, lda, 1, W, N, 1, Cm, N, bias, M, N, K};
    hipLaunchKernelGGL((k_gemm_f32<1, 1>), dim3((N + TBN - 1) / TBN, (M + TBM - 1) / TBM), dim3(256), 0, s, g);
}
// dA[M][K] = dZ[M][N] W[K][N]^T   (contraction over n; "B"(n, k) = W[k*N + n])
void gemm_nt(const float* dZ, const float* W, float* dA, int64_t ldda, int M, int N, int K, hipStream_t s) {
    GemmF32 g{dZ, N, 1, W, 1, N, dA, ldda, nullptr, M, K, N};
    hipLaunchKernelGGL((k_gemm_f32<1, 0>), dim3((K + TBN - 1) / TBN, (M + TBM - 1) / TBM), dim3(256), 0, s, g);
}
// dW[K][N] = A[M][K]^T dZ[M][N]   (contraction over rows; "A"(k, r) = A[r*lda + k])
void gemm_tn(const float* A, int64_t lda, const float* dZ, float* dW, int M, int N, int K, hipStream_t s) {
    GemmF32 g{A, 1, lda, dZ, N, 1, dW, N, nullptr, K, N, M};
    hipLaunchKernelGGL((k_gemm_f32<0, 1>), dim3((N + TBN - 1) / TBN, (K + TBM - 1) / TBM), dim3(256), 0, s, g);
}

int red_parts(int M) { return std::max(1, std::min(RED_PARTS, (M + 63) / 64)); }

}  // namespace

bool trainer_step(Trainer* t, const TrainHyper& h, const float* d_boards, const float* d_pis, const float* d_vs, int b,
                  uint64_t mask_seed, bool apply, hipStream_t s) {
    if (!t || b <= 1 || b > TRAIN_MAX_BATCH) return false;
    const int C = t->C;
    const Layout& L = t->L;
    float* P = t->params;
    float* G = t->grads;
    struct LayerDef { const float* A; int64_t lda; int M, K, N; int64_t w, bias, bn; bool drop; };
    const int rows[6] = {b * 42, b * 42, b * 20, b * 6, b, b};
    LayerDef ld[6] = {
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
        d.mask_seed = mask_seed;
        return d;
    };
    // ---- forward ----
    hipLaunchKernelGGL(k_boards_col1, grid1((int64_t)b * 42 * 20, 256, 1 << 20), dim3(256), 0, s, d_boards, t->col[0], b);
    for (int l = 0; l < 6; ++l) {
        const LayerDef& d = ld[l];
        gemm_nn(d.A, d.lda, P + d.w, t->z[l], P + d.bias, d.M, d.N, d.K, s);
        BnLayer bn = bn_desc(l, t->a[l], nullptr);
        const int parts = red_parts(d.M), rpb = (d.M + parts - 1) / parts;
        hipLaunchKernelGGL((k_colreduce<0>), dim3((d.N + 63) / 64, parts), dim3(256), 0, s, bn, t->z[l], rpb, t->partial);
        hipLaunchKernelGGL(k_bn_stats_finish, dim3((d.N + 255) / 256), dim3(256), 0, s, t->partial, parts, d.M, d.N, h.bn_eps,
                           h.bn_momentum, t->mean[l], t->invstd[l], P + d.bn + 2 * d.N, P + d.bn + 3 * d.N);
        hipLaunchKernelGGL(k_bn_apply, grid1((int64_t)d.M * d.N), dim3(256), 0, s, bn);
        if (l == 0) hipLaunchKernelGGL(k_im2col, grid1((int64_t)b * 42 * 9 * C / 4), dim3(256), 0, s, t->a[0], t->col[1], b, 6, 7, C, 1);
        if (l == 1) hipLaunchKernelGGL(k_im2col, grid1((int64_t)b * 20 * 9 * C / 4), dim3(256), 0, s, t->a[1], t->col[2], b, 6, 7, C, 0);
        if (l == 2) hipLaunchKernelGGL(k_im2col, grid1((int64_t)b * 6 * 9 * C / 4), dim3(256), 0, s, t->a[2], t->col[3], b, 4, 5, C, 0);
    }
    hipLaunchKernelGGL(k_heads_loss, dim3(b), dim3(64), 0, s, t->a[5], P + L.pi_w, P + L.pi_b, P + L.v_w, P + L.v_b, d_pis, d_vs, b,
                       t->dhead, t->sample_loss, t->logits);
    hipLaunchKernelGGL(k_loss_accumulate, dim3(1), dim3(64), 0, s, t->sample_loss, b, t->loss_totals);
    // ---- backward ----
    hipLaunchKernelGGL(k_heads_bwd, grid1(std::max<int64_t>(512 * 8 + 8, (int64_t)b * 512)), dim3(256), 0, s, t->a[5], t->dhead,
                       P + L.pi_w, P + L.v_w, b, G + L.pi_w, G + L.pi_b, G + L.v_w, G + L.v_b, t->dact);
    for (int l = 5; l >= 0; --l) {
        const LayerDef& d = ld[l];
        // t->dact holds d loss / d a[l]  ->  dz (through dropout, ReLU and BatchNorm)
        BnLayer bn = bn_desc(l, t->dz, t->dact);
        const int parts = red_parts(d.M), rpb = (d.M + parts - 1) / parts;
        hipLaunchKernelGGL((k_colreduce<1>), dim3((d.N + 63) / 64, parts), dim3(256), 0, s, bn, (const float*)nullptr, rpb, t->partial);
        hipLaunchKernelGGL(k_bn_bwd_finish, dim3((d.N + 255) / 256), dim3(256), 0, s, t->partial, parts, d.N, t->sums, G + d.bn,
                           G + d.bn + d.N);
        hipLaunchKernelGGL(k_bn_bwd_apply, grid1((int64_t)d.M * d.N), dim3(256), 0, s, bn, t->sums);
        // bias gradient = column sums of dz (zero up to rounding under BatchNorm; kept, as autograd keeps it)
        BnLayer cs{}; cs.M = d.M; cs.N = d.N;
        hipLaunchKernelGGL((k_colreduce<2>), dim3((d.N + 63) / 64, parts), dim3(256), 0, s, cs, t->dz, rpb, t->partial);
        hipLaunchKernelGGL(k_colsum_finish, dim3((d.N + 255) / 256), dim3(256), 0, s, t->partial, parts, d.N, G + d.bias);
        gemm_tn(d.A, d.lda, t->dz, G + d.w, d.M, d.N, d.K, s);
        if (l == 0) break;
        if (l >= 4) {
            gemm_nt(t->dz, P + d.w, t->dact, d.K, d.M, d.N, d.K, s);      // FC: d a[l-1] directly ([b][K])
        } else {
            gemm_nt(t->dz, P + d.w, t->dcol, d.K, d.M, d.N, d.K, s);
            const int H = l == 3 ? 4 : 6, W = l == 3 ? 5 : 7, pad = l == 1 ? 1 : 0;
            hipLaunchKernelGGL(k_col2im, grid1((int64_t)b * H * W * C / 4), dim3(256), 0, s, t->dcol, t->dact, b, H, W, C, pad);
        }
    }
    if (apply) {
        t->step += 1;
        const float bc1 = 1.0f - std::pow(h.beta1, (float)t->step), bc2 = 1.0f - std::pow(h.beta2, (float)t->step);
        hipLaunchKernelGGL(k_adam, grid1(L.total), dim3(256), 0, s, P, G, t->m, t->v, L.total, h.lr, h.beta1, h.beta2, h.adam_eps, bc1,
                           std::sqrt(bc2));
    }
    return hipGetLastError() == hipSuccess;
}

}  // namespace az
