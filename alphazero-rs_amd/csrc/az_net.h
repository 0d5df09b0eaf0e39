// az_net.h -- NNet::predict on the device (src/nnet.rs:35-45).
//
// Three nets answer a leaf batch: the reference's stub (examples/connect_four.rs:12-43),
// a deterministic test fixture, and the policy+value conv net of
// connect_four_net.py:20-95 as bf16 MFMA kernels (az_net.hip).
#pragma once
#include "az_tree.h"

namespace az {

struct ConvNet;        // weights of one model id, az_net.hip
struct NetWorkspace;   // activation workspace of one stream, az_net.hip

// offsets of the flat f32 parameter vector (the weights file)
struct Layout {
    int C;
    int64_t conv_w[4], conv_b[4], conv_bn[4], fc_w[2], fc_b[2], fc_bn[2], pi_w, pi_b, v_w, v_b, total;
    explicit Layout(int c) : C(c) {
        int64_t o = 0;
        for (int l = 0; l < 4; ++l) {
            int cin = l == 0 ? 2 : C;
            conv_w[l] = o; o += 9ll * cin * C;
            conv_b[l] = o; o += C;
            conv_bn[l] = o; o += 4ll * C;
        }
        const int fin[2] = {6 * C, 1024}, fout[2] = {1024, 512};
        for (int l = 0; l < 2; ++l) {
            fc_w[l] = o; o += (int64_t)fin[l] * fout[l];
            fc_b[l] = o; o += fout[l];
            fc_bn[l] = o; o += 4ll * fout[l];
        }
        pi_w = o; o += 512 * 7; pi_b = o; o += 7;
        v_w = o; o += 512; v_b = o; o += 1;
        total = o;
    }
};


struct NetProfile {          // filled when profiling is on; flops = what the matrix cores were asked to do (a layer that
    double conv2_ms = 0, conv2_flops = 0, total_ms = 0, total_flops = 0;   // runs as a table lookup contributes time, not flops)
    double conv3_ms = 0, conv3_flops = 0;
    double conv4_ms = 0, conv4_flops = 0;      // conv4 alone
    double fc_ms = 0, fc_flops = 0;            // fc1 + fc2 + heads
    double rows = 0;                           // executed rows of the timed forwards
    double conv2_bytes = 0;  // conv2 as a table: table rows gathered + activation rows written (algorithmic bytes)
    uint64_t launches = 0;
};

// pi = 1/7, v = +1 (kind 0) or the hash fixture (kind 1) for rows [0, *eb.n)
void launch_net_fixture(const EvalBatch& eb, int kind, uint64_t salt, hipStream_t s);

// weights of one model id
ConvNet* convnet_create(int channels, const char** err);
void convnet_destroy(ConvNet* n);
// activation workspace of one stream, shared by every model that runs on it
NetWorkspace* netws_create(int channels, int max_batch, const char** err);
void netws_destroy(NetWorkspace* ws);
int64_t convnet_param_count(int channels);
// raw f32 parameters (layout: DESIGN.md "weights file"); BN is folded and weights are
// rounded to bf16 on upload.
bool convnet_set_params(ConvNet* n, const float* host_params, int64_t count);
bool convnet_get_params(const ConvNet* n, float* host_params, int64_t count);
void convnet_init_random(ConvNet* n, uint64_t seed);
// fold finished profile records into *prof (call after the workspace's stream has been synchronised)
void netws_resolve_profile(NetWorkspace* ws, NetProfile* prof);
// k_conv3_auto's device-side accounting: adds (rows, working launches) since the last reset to out[2]; reset clears it
bool netws_conv3_accounting(NetWorkspace* ws, unsigned long long out[2], bool reset);
// Kernel-set switches of the conv net (az_set_option).  They live in the az_engine that was handed to az_set_option and are passed
// down with every forward: one engine's option never changes another engine's results.  The shipped library (built without
// -DAZ_DIAG) only honours conv2_table 0 / 1 and conv3_small 0 / 1; everything else selects superseded kernel generations, forced tiles,
// clock-stamp builds or timing ablations that compute WRONG results, which are compiled into libaz_engine_diag.so only (tools/).
struct NetOptions {
    int conv2_table = 1;    // 1: conv1 + conv2 as table gathers (k_conv2_table_x); 0: conv2 as the MFMA implicit GEMM (same function, other rounding)
    int conv3_small = 1;    // conv3 of a small expected batch on the 4-stage LDS-DMA ring (bit-identical)
    int conv3_planes = 1;   // conv3: the bank-conflict-free LDS image (Conv3Tables; bit-identical); 0 = image rows in order, 128 B each
    int ring_packed = 1;    // the LDS-DMA ring reads its weight stages from the model's packed copy (one stage = 16 KiB of consecutive bytes instead of
                            // 128 rows K * 2 bytes apart; conv4 -3 %; bit-identical); 0 = from the [N][K] weights
    int conv3_tail = 1;     // conv3: a short last round of workgroups is cut into half tiles (k_conv3_auto); 0 = full tiles only (bit-identical)
    int narrow_rows = 32;   // conv3 / conv4 / fc1 / fc2 of a batch of at most this many rows (x 2 for conv4, x 4 for the FCs) run as the register-fed
                            // skinny GEMM (k_gemm_skinny), decided on the device from the exact row count; 0 = never (bit-identical)
    // ---- diagnostic library only ----
    int conv3_pp = 0;       // 1: conv3 as the ping-pong kernel (k_conv3_pp: ONE 8-wave workgroup per CU, 12 boards x 256 channels, the two waves of a
                            // SIMD alternating LOAD and COMPUTE slots; bit-identical; measured in round 4, not faster: profiles/README.md); 16..40: its
                            // timing ablations / schedule variants (WRONG results)
    int gemm_variant = 5;   // 0 128x128 register-staged tiles everywhere; 1 / 2 256x256 LDS-DMA tiles; 3 conv2 image-resident, one 8-wave
                            // workgroup per CU; 5 the shipped set; 11-17 timing ablations of variant 2 (WRONG results)
    int conv1_table = 1;    // conv2 as a GEMM gathers its image from the conv1 table (1) / runs k_conv1 into act1 (0)
    int conv2_pipe = 1;     // conv2 as a GEMM: k_conv_same_pipe (1) / round 1's k_conv_img2 (0)
    int conv3_pipe = 1;     // conv3: 1 k_conv_valid_pipe with interleaved fragment reads; 2 without; 0 round 1's kernel; 3 clock stamps;
                            // 9-15 its timing ladder (WRONG results)
    int conv3_ring = 0;     // force conv3 onto the ring (1 / 2: 128-row tiles with 2 / 4 stages, 3: device-picked tile)
    int conv4_big = 0;      // conv4 on the 256x256 kernel (1 always, 2 from 4096 rows)
    int fc_ring = 1;        // 1 ring with the tile picked on the device; 2 picked on the host; 3 plain 128-row ring; 0 register-staged tiles
    int ring_tile[6] = {0, 0, 0, 0, 0, 0};   // per layer: forced BM * 10 + NS
};
// diagnostic variant 13 only: per-block {shader cycles, 100 MHz ticks} of the conv2 K loop
bool netws_read_clock_stamps(NetWorkspace* ws, unsigned long long* out2048);
// allocate what a forward under `opt` may need later (conv1's haloed image for the kernel sets that run conv1 as a kernel), so that
// convnet_forward never allocates -- it may run inside a stream capture
bool convnet_prepare(NetWorkspace* ws, const NetOptions& opt);
// forward for rows [0, *eb.n) of model n in workspace ws; n_rows_hint = host-side upper bound used to size the grids.
// If prof != nullptr the forward and its conv2 launch are bracketed with HIP events (resolved later).
// n_rows_typ = expected row count (kernel / tile choice only; 0 = n_rows_hint).
void convnet_forward(ConvNet* n, NetWorkspace* ws, const EvalBatch& eb, int n_rows_hint, int n_rows_typ, hipStream_t s, NetProfile* prof,
                     const NetOptions& opt);

}  // namespace az
