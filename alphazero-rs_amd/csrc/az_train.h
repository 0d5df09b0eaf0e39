// az_train.h -- NNet::train on the device (src/nnet.rs:38; recipe of connect_four_net.py:13-21, :102-151).
//
// f32 training of the policy+value net on the engine's flat parameter layout (az_net.hip `Layout`, the weights
// file): forward in BatchNorm training mode, softmax cross-entropy + mean squared error, backward, Adam.
// Every GEMM (conv layers as im2col x weights, dgrad, wgrad, the FCs) runs on the f32 matrix cores
// (v_mfma_f32_16x16x4_f32); the column reductions are two-stage and summed in a fixed order, so a step is
// deterministic run to run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace az {

struct Trainer;

struct TrainHyper {
    float lr = 1e-3f;            // connect_four_net.py:21
    float beta1 = 0.9f, beta2 = 0.999f, adam_eps = 1e-8f;
    float bn_momentum = 0.99f;   // tf.layers.batch_normalization default
    float bn_eps = 1e-3f;
    float dropout = 0.3f;        // connect_four_net.py:15
};

constexpr int TRAIN_MAX_BATCH = 256;

Trainer* trainer_create(int channels, const char** err);
void trainer_destroy(Trainer* t);
// host f32 parameters in weights-file order; resets the Adam moments and the step counter
bool trainer_set_params(Trainer* t, const float* host_params, int64_t count);
bool trainer_get_params(Trainer* t, float* host_params, int64_t count, hipStream_t s);
// One optimisation step on a device-resident batch: boards [b][2][6][7], pis [b][7], vs [b] (f32).
// mask_seed keys the dropout masks of this step (hash of (mask_seed, layer, element): az_common.h dropout_keep).
// apply = false computes the loss and the gradients only.  The per-step losses are accumulated on the device
// (trainer_read_losses).  Returns false on a HIP error.
bool trainer_step(Trainer* t, const TrainHyper& h, const float* d_boards, const float* d_pis, const float* d_vs, int b,
                  uint64_t mask_seed, bool apply, hipStream_t s);
// sums of (loss_pi, loss_v) over the steps since the last call with reset = true; synchronises the stream
bool trainer_read_losses(Trainer* t, double out[2], bool reset, hipStream_t s);
// gradients of the last step, weights-file order (running-stat slots are 0)
bool trainer_get_grads(Trainer* t, float* host_grads, int64_t count, hipStream_t s);
// device scratch for the caller's batch: [TRAIN_MAX_BATCH] x (84 + 7 + 1) floats
float* trainer_batch_boards(Trainer* t);
float* trainer_batch_pis(Trainer* t);
float* trainer_batch_vs(Trainer* t);
// One epoch of `steps` optimisation steps on device-resident samples: step i trains on rows d_idx[i*b .. i*b+b) of
// (all_boards [n][84], all_pis [n][7], all_vs [n]); its dropout masks are keyed by mix64(seed_key ^ (gstep0 + i)).
// The launch sequence of a step is captured in a hipGraph and replayed.  Synchronises the stream.
bool trainer_run_epoch(Trainer* t, const TrainHyper& h, const float* all_boards, const float* all_pis, const float* all_vs,
                       const int64_t* d_idx, int64_t steps, int b, uint64_t seed_key, uint64_t gstep0, hipStream_t s);
void trainer_set_graph(Trainer* t, bool on);      // A/B switch: replay a captured graph (default) or launch directly
void trainer_set_fwd_dma(Trainer* t, bool on);    // 1 (default): forward GEMMs fed by LDS-DMA (k_gemm_f32_dma); 0: register-staged k_gemm_f32
void trainer_set_fork(Trainer* t, bool on);       // true: a step's wgrad chains on a second stream branch, joined before Adam (default false: no gain)
void trainer_set_gemm3_ring(Trainer* t, bool on); // true (default): the big bf16 x 3 GEMMs on k_gemm3_ring (one 8-wave workgroup per CU, 3-stage ring)
void trainer_set_fwd_x3(Trainer* t, bool on);     // true (default): conv2..conv4 forward as f16 x 3 on the f16 matrix cores; false: v_mfma_f32_16x16x4_f32
void trainer_set_wgrad_tr(Trainer* t, bool on);   // true (default): conv wgrad on k_wgrad3_tr (transposed LDS reads; no transpose kernels)
void trainer_set_implicit(Trainer* t, bool on);   // true (default): conv2..conv4's GEMMs gather their rows from the activations (no im2col / col2im)
void trainer_set_gemm(Trainer* t, int mode);      // 1 (default): the GEMMs as bf16 x 3 on the bf16 matrix cores; 0: v_mfma_f32_16x16x4_f32

}  // namespace az
