/*
 * unet_hip.h - C ABI of the MI355X-native U-Net lane-segmentation path (libunet_hip.so).
 *
 * The reference has no FFI: its seam is the Python model container
 * `RKNN_model_container` (reference src/py_utils/rknn_executor.py:4-42), whose
 * `run()` hands a uint8 NHWC frame to a vendor runtime (`rknn.inference`,
 * rknn_executor.py:36) and gets the mask tensor back.  These entry points are
 * what a binding for that seam needs: plain pointers and sizes, no torch or
 * numpy types.  Each function names the reference interface it stands in for.
 *
 * Conventions
 *   - every function returns UNET_OK (0) or a non-zero unet_status code;
 *     unet_last_error() gives the text of the last failure on that handle
 *     (reference: `exit(ret)` on init failure, rknn_executor.py:16-18);
 *   - `*_dev` pointers are caller-owned DEVICE pointers on the handle's HIP
 *     device; `*_host` pointers are host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - the library owns only its packed-weight arena and its activation
 *     workspace; it never frees or reallocates caller memory;
 *   - calls on one handle are not re-entrant (the reference calls run() from a
 *     single rospy subscriber thread, src/unet_ros_node.py:280,313); they may
 *     come from any host thread (hipSetDevice is issued per call).
 */
#ifndef UNET_HIP_H
#define UNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct unet_ctx* unet_handle_t;

typedef enum unet_status {
  UNET_OK = 0,
  UNET_ERR_INVALID_ARG = 1, /* null pointer, bad enum, bad size */
  UNET_ERR_SHAPE = 2,       /* H or W not a multiple of 2^depth, numel mismatch */
  UNET_ERR_STATE = 3,       /* call order: params missing, not finalized, released */
  UNET_ERR_HIP = 4,         /* a HIP runtime call failed */
  UNET_ERR_NOMEM = 5,       /* device allocation failed */
  UNET_ERR_UNKNOWN_PARAM = 6,
  UNET_ERR_RANGE = 7        /* f16x3 tier: an activation left the fp16 range; re-run on the fp32 tier (unet_device_error) */
} unet_status;

#define UNET_MAX_DEPTH 6

/* Network description = constructor arguments of the reference's float model,
 * `UNet(in_channels, out_channels, features)` (reference README.md:1424). */
typedef struct unet_config {
  int32_t in_channels;              /* 3 */
  int32_t out_channels;             /* 1 (only 1 is supported by the head kernel) */
  int32_t depth;                    /* len(features), 1..UNET_MAX_DEPTH */
  int32_t features[UNET_MAX_DEPTH]; /* e.g. {64,128,256,512} */
  int32_t device;                   /* HIP device index (reference: device_id string, rknn_executor.py:5) */
  float input_mean[3];              /* per-channel (u8 - mean) / std, reference README.md:3110-3111 */
  float input_std[3];
} unet_config;

/* ---- lifecycle: stands in for RKNN() + load_rknn + init_runtime (rknn_executor.py:6-21) ---- */
int unet_create(const unet_config* cfg, unet_handle_t* out);

/* Hand over one state_dict tensor in PyTorch layout (fp32, host memory):
 * conv (O,I,3,3), ConvTranspose (I,O,2,2), BatchNorm vectors, head (1,C,1,1).
 * `name` is the reference state_dict key (README.md:1424-1447), e.g.
 * "encoder_blocks.0.0.weight".  The integer num_batches_tracked counters are not
 * parameters of the forward pass and are not passed.  May be called again later to
 * overwrite a tensor (followed by unet_finalize). */
int unet_load_param(unet_handle_t h, const char* name, const float* data_host, size_t numel);

/* Fold eval-mode BatchNorm into per-channel scale/shift, repack weights into
 * MFMA fragment order and upload them.  Fails with UNET_ERR_STATE if a
 * tensor of the network is missing. */
int unet_finalize(unet_handle_t h);

/* Number of state_dict float tensors the network expects / name of the i-th. */
int unet_num_params(unet_handle_t h);
const char* unet_param_name(unet_handle_t h, int index);
size_t unet_param_numel(unet_handle_t h, int index);

/* Activation workspace the library allocates for a batch shape (bytes);
 * unet_reserve allocates it ahead of time so forward does no hipMalloc. */
size_t unet_workspace_bytes(unet_handle_t h, int n, int height, int width);
int unet_reserve(unet_handle_t h, int n, int height, int width);

/* ---- the hot path: stands in for rknn.inference(inputs=[u8 NHWC]) (rknn_executor.py:36) ----
 * frames_dev : (N,H,W,3) uint8 RGB, un-normalised (reference src/unet.py:30-42)
 * logits_dev : (N,1,H,W) float32 pre-sigmoid logits, or NULL
 * probs_dev  : (N,1,H,W) float32 sigmoid(logits) - what the deployed blob returns
 *              (its last op is ConvSigmoid), or NULL
 * mask_dev   : (N,H,W) uint8, 255 where logit > threshold_logit else 0, or NULL
 *              (reference src/unet.py:67 thresholds the probability; callers
 *              pass threshold_logit = log(t/(1-t)), 0 for t = 0.5) */
int unet_forward_u8(unet_handle_t h, const uint8_t* frames_dev, int n, int height, int width,
                    float* logits_dev, float* probs_dev, uint8_t* mask_dev, float threshold_logit,
                    void* stream);

/* forward(image)->logits of the float model (reference README.md:1460-1481):
 * image_dev is (N,3,H,W) float32, already normalised, NCHW like the PyTorch module. */
int unet_forward_f32(unet_handle_t h, const float* image_nchw_dev, int n, int height, int width,
                     float* logits_dev, float* probs_dev, uint8_t* mask_dev, float threshold_logit,
                     void* stream);

/* bf16 tier of the same forward (BASELINE.json configs[2]): bf16 activations and weights, fp32 accumulate,
 * fp32 BatchNorm/ReLU epilogue, fp32 logits out.  A separate accuracy tier: the 1e-3 fp32 logit bound does not
 * apply.  Needs every feature width to be a multiple of 32. */
int unet_forward_u8_bf16(unet_handle_t h, const uint8_t* frames_dev, int n, int height, int width,
                         float* logits_dev, float* probs_dev, uint8_t* mask_dev, float threshold_logit,
                         void* stream);

/* Split-operand ("f16x3") tier of the same forward: every fp32 operand is carried as fp16 hi + lo and every
 * product is formed by three fp16 MFMAs with fp32 accumulation (csrc/conv_x3_ws.h).  Same accuracy class as the
 * exact-fp32 tier - it passes the fp32 parity tests (logits within 2e-4 of the reference's, masks identical off
 * ties) - at 3/16 of its MFMA cost.  Same arguments and outputs as unet_forward_u8 / unet_forward_f32.  Needs
 * in_channels == 3, every feature width a multiple of 64 (<= 512) and activations below 65504 in magnitude. */
int unet_forward_u8_x3(unet_handle_t h, const uint8_t* frames_dev, int n, int height, int width,
                       float* logits_dev, float* probs_dev, uint8_t* mask_dev, float threshold_logit,
                       void* stream);
int unet_forward_f32_x3(unet_handle_t h, const float* image_nchw_dev, int n, int height, int width,
                        float* logits_dev, float* probs_dev, uint8_t* mask_dev, float threshold_logit,
                        void* stream);

/* Release device memory: stands in for rknn.release() (rknn_executor.py:40-42).
 * Idempotent on a live handle pointer set to NULL by the caller; after it every
 * other call on the handle is invalid. */
int unet_destroy(unet_handle_t h);

/* ---- per-launch timing (measurement aid, stands in for the reference's wall-clock around run(),
 * src/unet.py:80-83): when enabled, every kernel launch of a forward call is bracketed by a
 * hipEvent pair on the caller's stream.  unet_profile_count synchronises on the recorded events. */
int unet_profile_enable(unet_handle_t h, int on);
int unet_profile_count(unet_handle_t h);
int unet_profile_get(unet_handle_t h, int index, char* name, size_t name_cap, double* ms, double* flops,
                     double* bytes);

/* ---- training step (reference README.md:2060-2084: zero_grad, model(images) in train mode,
 * BCEWithLogitsLoss README.md:1694-1709, loss.backward(), optimizer.step() README.md:2173) ----
 * Parameters, gradients, Adam moments and BatchNorm running statistics live in caller-owned flat
 * device buffers: tensors in PyTorch layout back to back, in unet_param_name() order, the
 * running_mean / running_var entries in `bn_buffers_dev`, everything else in the other four.
 * unet_train_layout gives each entry's offset (in floats) and which buffer it lives in. */
size_t unet_train_param_numel(unet_handle_t h);
size_t unet_train_buffer_numel(unet_handle_t h);
int unet_train_layout(unet_handle_t h, int index, int* is_buffer, size_t* offset);
int unet_train_attach(unet_handle_t h, float* params_dev, float* grads_dev, float* exp_avg_dev,
                      float* exp_avg_sq_dev, float* bn_buffers_dev);
size_t unet_train_workspace_bytes(unet_handle_t h, int n, int height, int width);

/* Loss of the step.  mode 0 (default): BCEWithLogitsLoss, mean (reference README.md:1694-1709).
 * mode 1: the reference training script's BCEDiceLoss (README.md:1855-1893, :2169-2170):
 *   bce_weight * BCEWithLogits(pos_weight) + dice_weight * (1 - (2 sum(s t) + smooth) / (sum s + sum t + smooth));
 * loss_dev then receives three floats {total, bce, dice}. */
int unet_train_set_loss(unet_handle_t h, int mode, float bce_weight, float dice_weight, float pos_weight,
                        float smooth);

/* Forward in train mode (batch statistics, running stats updated with momentum 0.1), mean
 * BCE-with-logits against targets_dev (N,1,H,W float 0/1), full backward.  Writes every parameter
 * gradient into grads_dev (overwriting: this is zero_grad + backward), the scalar loss into
 * loss_dev[0] (loss_dev must hold 4 floats; see unet_train_set_loss) and, if not NULL, the logits into logits_dev.  No communication: a data-parallel
 * caller all-reduces grads_dev between this call and unet_train_adam_step. */
int unet_train_forward_backward_u8(unet_handle_t h, const uint8_t* frames_dev, const float* targets_dev, int n,
                                   int height, int width, float* loss_dev, float* logits_dev, void* stream);
int unet_train_forward_backward_f32(unet_handle_t h, const float* image_nchw_dev, const float* targets_dev, int n,
                                    int height, int width, float* loss_dev, float* logits_dev, void* stream);

/* torch.optim.Adam (decoupled = 0) or AdamW (decoupled = 1) on the flat buffers, `step` counted
 * from 1; gradients are multiplied by grad_scale first (1/world_size after a SUM all-reduce).
 * Re-derives the packed MFMA operands from the updated parameters. */
int unet_train_adam_step(unet_handle_t h, int step, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int decoupled, float grad_scale, void* stream);

/* Data-parallel overlap (north_star: RCCL all-reduce of the gradients; the reference has no distributed code).  The
 * backward pass finishes the gradients of the decoder, the bottleneck and the head - the tail
 * [unet_train_grad_split, unet_train_param_numel) of the flat gradient buffer - before it starts on the encoder.  With a
 * communication stream set, unet_train_forward_backward_* makes that stream wait for the point where the tail is final,
 * so the caller's all-reduce of the tail, enqueued on that stream, overlaps the encoder's backward. */
int unet_train_set_comm_stream(unet_handle_t h, void* comm_stream);
size_t unet_train_grad_split(unet_handle_t h);

/* Process-wide switch for the training step's 3x3 convolutions: 1 (default) = forward, input gradient AND weight
 * gradient on the split-operand fp16 kernels (csrc/conv_x3_ws.h, conv_x3_r512.h, wgrad_x3_ws.h: fp16 hi + lo operands,
 * three MFMAs per product, fp32 accumulate) wherever Cin and Cout are multiples of 64 (environment
 * UNET_TRAIN_X3_WGRAD=0 keeps only the weight gradients on the exact-fp32 kernels); 0 = exact-fp32 MFMA kernels
 * everywhere.  BatchNorm and the loss are fp32 either way.  Accuracy: products carry ~2^-22 relative error while the
 * operands' lo parts are normal fp16 numbers; the training weight packs are not pre-scaled per channel (they are
 * re-derived on the device after every optimizer step), so for |w| < 2^-3 the lo part is subnormal and the product
 * error becomes an absolute ~2^-25 - inside the gradient tolerances of tests/test_train_gpu.py, which run in both
 * modes.  Returns the previous setting; environment UNET_TRAIN_X3=0 sets the initial value to 0. */
int unet_set_train_x3(int on);

/* Process-wide switch for WHERE the training step's f16x3 weight-gradient kernels run (no reference counterpart: the
 * reference's backward is torch autograd, README.md:2198-2201, which orders nothing beyond data dependence either).
 * 0 = in line on the caller's stream; 1 (default) = on a second, lower-priority stream owned by the handle, forked as
 * soon as the unit's dZ exists; 2 = forked behind the unit's input-gradient convolution, so that the weight gradient
 * runs beside the next unit's HBM-bound BatchNorm backward.  The side stream is joined back into the caller's stream before
 * the late-gradient event of unet_train_set_comm_stream and before unet_train_forward_backward_* returns its work to
 * the stream, so callers see no difference in ordering; results are bit-identical in all modes.  Ignored (in line) while
 * per-launch profiling or a debug snapshot is active.  mode outside 0..2 only queries.  Returns the previous mode;
 * environment UNET_TRAIN_SIDE sets the initial value. */
int unet_set_train_side(int mode);

/* Re-derive the packed MFMA operands from the attached parameter buffer after the caller overwrote it
 * (checkpoint load: reference README.md:2231 `model.load_state_dict`).  Unlike a second unet_train_attach it keeps
 * the loss configuration (unet_train_set_loss) and the workspace.  Synchronises the stream. */
int unet_train_repack(unet_handle_t h, void* stream);

/* Dice metric of the reference's validation loop (README.md:2115-2120 `compute_dice`, called at :2103-2104 with
 * pred = sigmoid(outputs) > 0.5): out_dev[0] = (2 sum(pred t) + smooth) / (sum pred + sum t + smooth) with
 * pred = logit > threshold_logit; out_dev[1..3] = the three sums.  logits/targets: `numel` floats each. */
int unet_dice_metric(int device, const float* logits_dev, const float* targets_dev, size_t numel,
                     float threshold_logit, float smooth, float* out_dev, void* stream);

const char* unet_last_error(unet_handle_t h);
const char* unet_version(void);

/* Asynchronous kernel-side conditions, kept in a per-handle error block the kernels write to.
 *  - A kernel whose bounded wave-progress wait gives up (csrc/wino_f32.h) records it instead of continuing silently
 *    with stale data.  The next unet_forward_* / unet_train_forward_backward_* call that sees the record returns
 *    UNET_ERR_HIP (like an asynchronous HIP error the failing launch may be an earlier one) and clears it, so one
 *    transient failure is reported once.
 *  - The f16x3 tier stores activations as fp16 hi + lo planes: |v| <= 65504, where the reference's fp32 network
 *    (README.md:1449-1458) has no such bound.  Activations are stored scaled by a per-channel power of two chosen
 *    from the BatchNorm parameters so that four standard deviations sit near 2^10 (csrc/unet_x3.inc), which leaves
 *    the range only for inputs hundreds of standard deviations from the running statistics; a kernel that meets such a
 *    value records it.  Only unet_device_error reports that (UNET_ERR_RANGE): the results of the calls since the last
 *    unet_device_error are then not at fp32 parity and the caller re-runs them on the fp32 tier
 *    (py_utils/rknn_executor.py does).
 * unet_device_error synchronises the device, returns UNET_ERR_HIP / UNET_ERR_RANGE / UNET_OK for everything launched on
 * this handle since its last call, and clears the block.  The reference's container has no equivalent:
 * rknn.inference reports failure through its return value (rknn_executor.py:36). */
int unet_device_error(unet_handle_t h);
/* The same for a caller that launched everything on one stream: waits for that stream only (other streams of the
 * device - a camera stage, a second model - keep running), then reports and clears as unet_device_error does. */
int unet_device_error_on(unet_handle_t h, void* stream);
/* Data-parallel callers (trainer.py; the reference has no distributed code, BASELINE.json north_star asks for an RCCL
 * all-reduce of the gradients): enqueue on `stream` a one-thread kernel that writes 1.0f to the device float `dst` if the
 * error block holds any record of the launches before it on that stream, else 0.0f.  Nothing is synchronised or cleared.
 * The trainer puts `dst` in front of its flat gradient bucket, so the word is summed over the ranks by the SAME
 * all-reduce as the gradients and every rank learns whether ANY rank failed before any of them updates its parameters. */
int unet_device_status_to(unet_handle_t h, float* dst, void* stream);

/* Test hook: write `value` into word `word` (0 = kernel failure, 1 = fp16 range) of the handle's error block, as a
 * kernel would.  Lets the host-side recovery paths be exercised without a failing kernel. */
int unet_debug_set_error_block(unet_handle_t h, int word, unsigned value);
/* Test hook, host arithmetic only: the power-of-two scale the f16x3 tier stores a BatchNorm channel's activations with
 * (csrc/unet_x3.inc, ActScale): 4 |gamma| + |beta| lands in [512, 1024), the scale clamped to [2^-40, 2^40], 1 for a
 * channel whose magnitude is below 1e-30. */
float unet_debug_act_scale(float gamma, float beta);

/* Process-wide algorithm switch for 3x3 convolutions with Cin % 16 == 0 on even-sized maps:
 * 1 = Winograd F(2x2,3x3) on the fp32 MFMA pipe (default), 0 = direct implicit GEMM.
 * Returns the previous setting.  Environment UNET_NO_WINOGRAD=1 sets the initial value to 0. */
int unet_set_winograd(int on);

/* Process-wide kernel choice for the bf16 tier's 3x3 convolutions.  -1 = automatic (default): the wave-specialised
 * kernel (csrc/conv_bf16_ws.h) on wide maps with enough tiles per CU, the one-wave-per-SIMD kernel
 * (csrc/conv_bf16_r512.h) on maps whose width is a multiple of 28 (or 14) with Cout % 128 == 0 and a work item for
 * half the CUs, the 2x2-wave kernel (csrc/igemm_bf16.h) otherwise; 0 = the 2x2-wave kernel only; 1 = the
 * wave-specialised kernel whenever the layer shape allows it (Cin % 64 == 0, Cout % 64 == 0 and <= 512, H % 16 == 0);
 * 2 = the one-wave-per-SIMD kernel whenever the shape allows it.  All three accumulate chunk by chunk, tap by tap and
 * give bit-identical results (tests/test_bf16_gpu.py).  The tier's ConvTranspose2d follows the same switch: mode 1 its
 * wave-specialised kernel (csrc/upconv_bf16_ws.h), mode 2 - and automatically, once there is a work item for half of
 * the CUs - its one-wave-per-SIMD kernel (csrc/upconv_bf16_r512.h; Cin % 128 == 0).  Returns the previous setting. */
int unet_set_bf16_persistent(int mode);

/* ---- single operators, for parity tests against the oracle (tests/test_ops_gpu.py) ----
 * All tensors are dense NHWC float32 device buffers.  Weights are passed in
 * PyTorch layout on the HOST and packed internally (slow path, test only). */

/* y = relu?(conv3x3(x, w) * scale + shift), pad 1, stride 1 (reference README.md:1452-1457).
 * x (N,H,W,Cin) -> y (N,H,W,Cout); w_host (Cout,Cin,3,3); scale/shift host (Cout). */
int unet_op_conv3x3(int device, const float* x_dev, int n, int h, int w, int cin, const float* w_host,
                    const float* scale_host, const float* shift_host, int cout, int relu,
                    float* y_dev, void* stream);

/* ConvTranspose2d k=2 s=2 with bias (reference README.md:1442): x (N,H,W,Cin) -> y (N,2H,2W,Cout);
 * w_host (Cin,Cout,2,2). */
int unet_op_upconv2x2(int device, const float* x_dev, int n, int h, int w, int cin, const float* w_host,
                      const float* bias_host, int cout, float* y_dev, void* stream);

/* Plain 1x1 convolution without bias (the GEMM behind the ConvTranspose2d input gradient):
 * x (N,H,W,Cin) -> y (N,H,W,Cout); w_host (Cout,Cin). */
int unet_op_conv1x1(int device, const float* x_dev, int n, int h, int w, int cin, const float* w_host, int cout,
                    float* y_dev, void* stream);

/* MaxPool2d(2,2) (reference README.md:1429): x (N,H,W,C) -> y (N,H/2,W/2,C). */
int unet_op_maxpool2x2(int device, const float* x_dev, int n, int h, int w, int c, float* y_dev, void* stream);

/* 1x1 head with bias (reference README.md:1447): x (N,H,W,C) -> logits (N,H,W). */
int unet_op_head1x1(int device, const float* x_dev, int n, int h, int w, int c, const float* w_host,
                    float bias, float* logits_dev, void* stream);

/* The same two operators in the split-operand tier (fp32 NHWC in and out; converted to / from fp16 hi + lo planes
 * internally).  cin, cout multiples of 64.  tile_width: 0 = automatic; 16 or 32 = force that pixel-tile shape of the
 * first kernel structure (csrc/conv_x3_ws.h); 28 (W % 28 == 0) or 14 (W == 14) = force the second structure
 * (csrc/conv_x3_r512.h; cout a multiple of 128; + 200 = its two-waves-along-the-pixels form even where cout is a
 * multiple of 256); 332 / 316 / 308 = the second structure's 7 x 32 / 14 x 16 / 28 x 8 pixel tiles (W a multiple of
 * 32 / 16 / 8, cout a multiple of 256); 532 = the 7 x 32 tile in the two-waves-along-the-pixels form (cout a multiple
 * of 128); 428 / 414 = csrc/conv_q8_r512.h (see unet_set_x3_cross_fp8); 628 / 632 = the third structure
 * (csrc/conv_x3_t448.h: 16 x 28 / 16 x 32 pixel tiles, W a multiple of 28 / 32, any cout that is a multiple of 64; the
 * network takes it by itself for the layers whose cout is 64 or 128); UNET_ERR_HIP if the forced structure does not
 * support the shape.
 * y_pool_dev: optional (N,H/2,W/2,Cout) MaxPool2d(2,2) output (reference README.md:1429). */
int unet_op_conv3x3_x3(int device, const float* x_dev, int n, int h, int w, int cin, const float* w_host,
                       const float* scale_host, const float* shift_host, int cout, int relu, int tile_width,
                       float* y_dev, float* y_pool_dev, void* stream);
/* The network's last two layers as the split-operand tier runs them (reference README.md:1456-1457, :1447, :1481): a
 * 3x3 convolution to 64 channels + scale/shift (+ ReLU) with the 1x1 head (64 -> 1, bias) fused into its epilogue; the
 * 64-channel activation is never stored.  x (N,H,W,cin) fp32 -> logits (N,H,W).  tile_width 0 / 16 / 32: first
 * structure; 628 / 632: third structure. */
int unet_op_conv3x3_x3_head(int device, const float* x_dev, int n, int h, int w, int cin, const float* w_host,
                            const float* scale_host, const float* shift_host, int relu, int tile_width,
                            const float* head_w_host, float head_bias, float* logits_dev, void* stream);
int unet_op_upconv2x2_x3(int device, const float* x_dev, int n, int h, int w, int cin, const float* w_host,
                         const float* bias_host, int cout, float* y_dev, void* stream);

/* Which kernel structure the split-operand tier's ConvTranspose2d (and the plain GEMMs of its training path) run on
 * (reference README.md:1442, :1476): -1 = automatic - the one-wave-per-SIMD kernel (csrc/upconv_x3_r512.h: 224-pixel
 * tiles, weights straight from L2) where Cin % 128 == 0 and there is a work item for at least half of the CUs, the
 * wave-specialised kernel (csrc/upconv_x3_ws.h) otherwise; 0 = the wave-specialised kernel only; 1 = the
 * one-wave-per-SIMD kernel whenever Cin % 128 == 0.  Both accumulate chunk by chunk in the same order and give
 * bit-identical results (tests/test_x3_gpu.py).  Returns the previous setting. */
int unet_set_x3_upconv_r512(int mode);

/* "f16q8": the split-operand tier with the two cross terms of every product (w_lo x_hi + w_hi x_lo, 2^-11 of the product)
 * formed on the fp8 matrix pipe (csrc/conv_q8_r512.h) in the 3x3 convolutions that suit it (Cin % 64 == 0,
 * Cout % 256 == 0, map width a multiple of 28 or 14, a work item for half of the CUs); the main term stays fp16 and
 * every other layer runs as in the f16x3 tier.  An accuracy tier of its own: logits within BASELINE.json's 1e-3 of the
 * reference's (measured on an MI355X, batch 256: 7.8e-4 against the reference's golden logits, 9.2e-4 against the f16x3
 * tier over 256 random frames - a 1.2x margin; binary masks differ from the f16x3 tier's in ~7 pixels per million, all at
 * logits within 1e-3 of zero), not the 2e-4 the f16x3 tier is held to (reference README.md:1449-1458 is plain fp32).
 * 0 = off (default), 1 = on for the following unet_forward_*_x3 calls of the CALLING THREAD (the switch is thread-local, so
 * a caller that sets it around one forward and restores it - UNetHIP.run_u8(precision="f16q8") - cannot change what another
 * thread's forward computes); the first call after switching it on rebuilds the handle's operators with the extra weight
 * fragments (it synchronises the device: do not mix with HIP graphs captured from the same handle).  Returns the previous
 * setting.
 * unet_op_conv3x3_x3 runs the kernel directly with tile_width 428 (W % 28 == 0) / 414 (W == 14). */
int unet_set_x3_cross_fp8(int mode);

/* Debug aid: during the next unet_train_forward_backward_* calls copy one internal buffer to dst_dev
 * (at most max_floats).  stage = 100+j: gradient w.r.t. the input of decoder step j's ConvTranspose2d;
 * 200+j: its space-to-depth gradient; 300+u / 400+u / 500+u: dZ, z and the saved BatchNorm statistics
 * of conv unit u (encoder, bottleneck, decoder order); -1 disables. */
int unet_train_debug_snapshot(unet_handle_t h, int stage, float* dst_dev, size_t max_floats);

/* dW of a 3x3 convolution (training backward): dz (N,H,W,Cout), x (N,H,W,Cin) -> dw_dev (Cout,Cin,3,3). */
int unet_op_wgrad3x3(int device, const float* dz_dev, const float* x_dev, int n, int h, int w, int cin, int cout,
                     float* dw_dev, void* stream);
/* The same weight gradient on the split-operand fp16 kernel (every fp32 operand as fp16 hi + lo, three MFMAs per
 * product, pixels as the GEMM's K dimension through transposed LDS reads); cin, cout multiples of 64, h even.
 * scaled != 0: dz is brought into the fp16 range by a power of two first, as the training step does for gradients
 * (`loss.backward()` of README.md:2077 computes these sums in fp32). */
int unet_op_wgrad3x3_x3(int device, const float* dz_dev, const float* x_dev, int n, int h, int w, int cin, int cout,
                        float* dw_dev, int scaled, void* stream);

/* ---- int8 tier of the deployed network ("model B") --------------------------------------
 * Stands in for the quantised .rknn blob behind rknn.inference (src/py_utils/rknn_executor.py:36): per-tensor
 * asymmetric int8 activations, per-output-channel asymmetric int8 weights as the reference configures its conversion
 * (README.md:3106-3116 `asymmetric_quantized-8` / `channel`; README.md:3370-3383), BatchNorm folded, sigmoid head, on
 * v_mfma_i32_16x16x64_i8.  The quantised model is a set of named arrays produced by unet_lane_detection_amd/quant.py
 * (int8 `<unit>.w_q`, int32 `<unit>.w_zp` / `.bias_q` / `.x_zp` / `.y_zp` / `.relu`, float32 `<unit>.mult`, the input
 * table `input.lut` int8[3][256] + `input.zp`; <unit> = the reference's state_dict prefixes such as
 * "encoder_blocks.0.0", "decoder_blocks.0", "output").  Integer-exact against oracle/int8_oracle.py; parity with the
 * Rockchip runtime itself is unpinned (its blobs cannot be executed here). */
typedef struct unet_i8_ctx* unet_i8_handle_t;
int unet_i8_create(int depth, const int* features, int device, unet_i8_handle_t* out);
int unet_i8_load(unet_i8_handle_t h, const char* name, const void* data_host, size_t bytes);
int unet_i8_finalize(unet_i8_handle_t h);
/* frames (N,H,W,3) uint8 as unet_forward_u8; logits = float32(int32 accumulator) * (x_scale * w_scale), probabilities
 * = sigmoid(logits) (what the blob's ConvSigmoid returns), mask as unet_forward_u8. */
int unet_i8_forward_u8(unet_i8_handle_t h, const uint8_t* frames_dev, int n, int height, int width, float* logits_dev,
                       float* probs_dev, uint8_t* mask_dev, float threshold_logit, void* stream);
/* Parity aid: copy one int8 activation tensor of the last forward to the host, dense NHWC with its real channels.
 * name: "im2col", "enc<l>.a", "cat<l>", "cat<l>.pool", "bott.a", "bott.b", "dec<j>.a", "dec<j>.b". */
int unet_i8_read_tensor(unet_i8_handle_t h, const char* name, int8_t* dst_host, size_t cap_bytes, int* channels);
int unet_i8_destroy(unet_i8_handle_t h);
const char* unet_i8_last_error(unet_i8_handle_t h);

/* Calibration pass on the FLOAT handle (README.md:3046-3078: min/max over calibration frames, algorithm 'normal'):
 * runs unet_forward_u8 and reports (min, max) of every activation tensor that carries quantisation parameters,
 * 2 floats per tensor in the order of quant.tensor_names(): input, per level {first encoder conv, concat tensor},
 * the two bottleneck convs, per decoder step its two convs.  Synchronises the stream. */
int unet_num_range_tensors(unet_handle_t h);
int unet_forward_u8_ranges(unet_handle_t h, const uint8_t* frames_dev, int n, int height, int width,
                           float* ranges_host, void* stream);

/* ---- camera stage on the GPU ----------------------------------------------------
 * Replaces the OpenCV calls of the reference's ROS callback (src/unet_ros_node.py:296-311) and of
 * RKNNLaneInference.preprocess_image / postprocess_output (src/unet.py:33, :70).  Integer arithmetic restated from
 * OpenCV 4.x's 8-bit paths; parity against cv2 itself is unpinned (cv2 is not installed and the reference has no
 * fixture for this stage) - see oracle/camera_oracle.py.
 *
 * unet_ipm_prestage_u8: img_dev = the data field of a sensor_msgs/Image with encoding bgr8 (bgr_in = 1) or rgb8
 * (bgr_in = 0): `height` rows of `step` bytes.  Computes cv2.warpPerspective(img, M, (warp_w, warp_h)) [INTER_LINEAR,
 * constant border 0], the reference's same-size INTER_AREA resize (a copy), the conversion to RGB and
 * cv2.resize(.., (out_w, out_h)) [bilinear] in one pass; minv = M^-1 (row-major 3x3, destination -> source).
 * out_rgb_dev: (out_h, out_w, 3) uint8, ready for unet_forward_u8. */
int unet_ipm_prestage_u8(int device, const uint8_t* img_dev, int height, int width, int step, int bgr_in,
                         const double minv[9], int warp_w, int warp_h, int out_w, int out_h, uint8_t* out_rgb_dev,
                         void* stream);

/* cv2.resize(src, (out_w, out_h)) [bilinear] of an 8-bit image with `channels` interleaved channels: the mask's way
 * back to the warped size (src/unet.py:70). */
int unet_resize_u8(int device, const uint8_t* src_dev, int height, int width, int channels, int out_w, int out_h,
                   uint8_t* dst_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNET_HIP_H */
