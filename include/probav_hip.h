/* probav_hip.h -- C ABI of libprobav_hip.so: the gfx950 (MI355X) implementation of the WDSR-B Conv3D
 * forward/backward hot path of mmbajo/PROBA-V and its shift-compensated loss.
 *
 * The reference has no native/FFI layer: the path sits behind Python callables that dispatch to
 * TensorFlow ops.  Each entry point below names the reference call it replaces
 * (paths relative to the reference repository).  A binding needs nothing but this header: plain
 * pointers to DEVICE memory (fp32 unless stated), sizes, and a hipStream_t passed as void*.
 *
 * Conventions
 *   - activations [N][H][W][T][C], C innermost (the reference's layout; models/modelsTF.py:19);
 *     kernels [kh][kw][kt][Cin][Cout] (Keras).  All buffers contiguous, 16-byte aligned.
 *   - every call only ENQUEUES work on `stream` and returns; it never synchronises, allocates or
 *     frees device memory (so calls can be captured in a hipGraph).  probav_engine_create /
 *     _destroy are the only functions that allocate (a few KB for the layer table).
 *     probav_forward / probav_backward additionally fork work that is off the critical path (the
 *     low-frequency residual path, the sums of the backward-filter slabs) onto one engine-owned side
 *     stream by event and join it back into `stream` before returning: the caller sees plain
 *     stream order.  The side stream is created on the engine's first pass (make that pass outside
 *     a graph capture); PROBAV_NO_SIDE_STREAM=1 in the environment disables it.
 *   - return value: 0 = ok, PROBAV_EINVAL (-1) bad argument, PROBAV_ENOSPACE (-2) workspace too small,
 *     PROBAV_EHIP (-3) a HIP call failed; probav_last_error() gives the text (thread-local).
 *   - thread-safety: an engine handle and its workspace may be used by one thread at a time;
 *     distinct handles are independent (one process per GPU under data parallelism).
 */
#ifndef PROBAV_HIP_H
#define PROBAV_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PROBAV_ABI_VERSION 7

/* Hyper-parameters of WDSRConv3D(name, band, mean, std, maxShift).build(scale, numFilters, kernelSize=3,
 * numResBlocks, expRate, decayRate, numImgLR, patchSizeLR, isGrayScale)   (models/modelsTF.py:8-17) */
typedef struct probav_net_cfg {
    int32_t scale;            /* 3 */
    int32_t num_filters;      /* 32 */
    int32_t num_res_blocks;   /* 12 */
    int32_t exp_rate;         /* 8 */
    int32_t dec_channels;     /* int(numFilters*decayRate) = 25 (models/modelsTF.py:182) */
    int32_t num_img_lr;       /* 9 (7, 13 also defined by the reference; models/modelsTF.py:62-69) */
    int32_t patch_size_lr;    /* 16 */
    int32_t max_shift;        /* 6 */
    float mean, std;          /* per-band constants (train.py:47-52) */
    int32_t in_channels;      /* 1: isGrayScale=True (every shipped cfg); 3: isGrayScale=False -- the input is [N,H,W,T,3], mainConv1 and
                                 residConv1 take three channels, the output stays one channel (models/modelsTF.py:19-20, :23-27) */
} probav_net_cfg;

typedef struct probav_engine probav_engine;

int probav_abi_version(void);
const char* probav_last_error(void);

/* ---- engine: the whole network ------------------------------------------------------------------ */
/* replaces WDSRConv3D(...).build(...)                                   models/modelsTF.py:15-43     */
int probav_engine_create(const probav_net_cfg* cfg, probav_engine** out);
void probav_engine_destroy(probav_engine* e);
/* number of trainable fp32 parameters (535 267 for p16t9c85r12) and the flat layout: layers in Keras
 * checkpoint order, per layer [g(Cout) | v(taps*Cin*Cout) | bias(Cout)]               */
int64_t probav_param_count(const probav_engine* e);
int probav_num_layers(const probav_engine* e);
/* name, offsets (in floats) and kernel shape of layer i; shape is [kh,kw,kt,Cin,Cout]               */
int probav_layer_info(const probav_engine* e, int i, char name[32], int64_t* g_off, int64_t* v_off,
                      int64_t* b_off, int32_t shape[5]);
/* kernel family: 0 = generic direct (VALU) kernels everywhere, 1 = fp32-MFMA row-tile kernels, 2 = fp32 MFMA + strip convolution,
 * 3 = 2 with the x6 kernels where they exist: fp32 in / fp32 out / fp32 accumulate, every fp32 product evaluated as six exact
 * bf16-piece products on the bf16 MFMA pipe, 4 (default) = the same kernels with the H3 arithmetic: three exact products of fp16 piece
 * pairs, every operand scaled by a power of two chosen from the largest magnitude of its scaling group -- one SAMPLE of an activation /
 * gradient tensor, one output COLUMN of a filter matrix (amax slots in the workspace, filled by the producing kernels).  3 and 4 are
 * held to the tolerances of 2 in tests/test_gpu_parity.py.  Results of one family are bitwise reproducible run to run; different
 * families differ by fp32 rounding.  In every family the forward result of a sample is independent of its batch mates, bit for bit
 * (models/modelsTF.py:15-43 has no cross-sample term).                                                                            */
int probav_engine_set_impl(probav_engine* e, int impl);
size_t probav_workspace_bytes(const probav_engine* e, int batch, int training);
/* per-kernel-class timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 * probav_engine_profile(e, 1, n) creates events for n launches and starts recording (allocation
 * happens here, never inside forward/backward); after a stream synchronise,
 * probav_engine_profile_read sums elapsed ms, algorithmic MACs and launch counts per class
 * (nclass >= 8: wn, small, conv3 fwd, conv3 bwd-data, conv3 wgrad, 1x1x1 fwd, 1x1x1 bwd-data,
 * 1x1x1 wgrad) and clears the log.                                                                  */
int probav_engine_profile(probav_engine* e, int enable, int max_launches);
/* restrict the bracketing to the kernel classes whose bit is set (default: all).  An event pair around EVERY launch costs ~6 % of a
 * training step (the launches no longer overlap their ramps); bench.py brackets only the dominant class inside its timed region.  */
int probav_engine_profile_classes(probav_engine* e, uint32_t mask);
int probav_engine_profile_read(probav_engine* e, int nclass, double* ms, double* macs, int64_t* launches);

/* replaces  model(x, training=...)      models/trainClass.py:127,139 ; test.py:117 ; testClass.py:26
 * x [B, P+maxShift, P+maxShift, T, 1] -> y [B, scale*P, scale*P, 1].  With training != 0 the
 * activations needed by probav_backward stay in `ws`.                                               */
int probav_forward(probav_engine* e, const float* params, const float* x, float* y, void* ws,
                   size_t ws_bytes, int batch, int training, void* stream);
/* replaces  tape.gradient(loss, model.trainable_variables)              models/trainClass.py:131
 * dy [B, scale*P, scale*P, 1] -> grads[param_count] (overwritten).  Must follow probav_forward
 * (training=1) on the same ws / batch.                                                              */
int probav_backward(probav_engine* e, const float* params, const float* dy, float* grads, void* ws,
                    size_t ws_bytes, int batch, void* stream);
/* The same in two buffers (ABI 5).  probav_workspace_bytes(e, batch, 1) = saved_bytes + scratch_bytes: the first part is the SAVED STATE of a
 * training forward pass (what `tf.GradientTape` keeps: models/trainClass.py:126-131) -- probav_forward(training=1) needs no more than that --,
 * the second is what only the reverse pass writes on its way (gradient buffers, partial-sum slabs, its amax slots; meaningless before and
 * after).  probav_backward_split READS `saved` and writes `scratch`: the saved state of one forward pass can serve any number of reverse
 * passes (a retained graph, a gradient check), and a framework that tracks which operator writes which of its arguments sees a functional
 * operator.  wcache: the weight cache the forward pass ran from, or NULL.  probav_backward(ws) = probav_backward_split(ws, ws + saved_bytes). */
int probav_workspace_split(const probav_engine* e, int batch, size_t* saved_bytes, size_t* scratch_bytes);
int probav_backward_split(probav_engine* e, const float* params, const float* dy, float* grads, const void* saved, size_t saved_bytes,
                          void* scratch, size_t scratch_bytes, int batch, const void* wcache, size_t wcache_bytes, void* stream);

/* ---- optimizer update fused with the weight normalisation of the next step ---------------------------------
 * replaces  optimizer.apply_gradients(...)  +  the WeightNormalization kernel recomputation of the NEXT model call
 *           (models/trainClass.py:132 ; models/modelsTF.py:191-197 ; Keras Nadam / Adam / SGD of train.py:77-83 by coefficients, as probav_nadam_step)
 * One launch updates all parameters in place (params, m, v: probav_param_count floats) and writes the effective weights of the UPDATED
 * parameters (both layouts, inverse norms, amax slots); a second one packs the MFMA operand fragments.  Everything lands in the
 * caller-owned weight cache (probav_weight_cache_bytes).  probav_forward_wc / probav_backward_wc are probav_forward / probav_backward
 * reading that cache instead of recomputing it: three launches leave every training step.  The cache is valid exactly as long as
 * `params` is not modified by anyone else.                                                                                          */
size_t probav_weight_cache_bytes(const probav_engine* e);
int probav_optimizer_step_fused(probav_engine* e, float* params, const float* grads, float* m, float* v, float lr, float beta1,
                                float beta2, float eps, float c_g, float c_m, float c_v, void* wcache, size_t wcache_bytes, void* stream);
/* The same cache from parameters that nothing is updating (inference, evaluation: `model(x)` many times on fixed weights -- test.py:117,
 * models/testClass.py:26): weight normalisation + operand packing once, then every probav_forward_wc starts at its first convolution. */
int probav_weight_cache_build(probav_engine* e, const float* params, void* wcache, size_t wcache_bytes, void* stream);
int probav_forward_wc(probav_engine* e, const float* params, const float* x, float* y, void* ws, size_t ws_bytes, int batch,
                      int training, const void* wcache, size_t wcache_bytes, void* stream);
int probav_backward_wc(probav_engine* e, const float* params, const float* dy, float* grads, void* ws, size_t ws_bytes, int batch,
                       const void* wcache, size_t wcache_bytes, void* stream);

/* ---- loss / metric ------------------------------------------------------------------------------ */
/* replaces Losses.shiftCompensatedL1Loss / L2Loss / cPSNR               models/loss.py:37-84
 * hr, pred [B,S,S,1] f32; mask [B,S,S,1] uint8 (non-zero = clear pixel).  Outputs (device):
 * l1[B], l2[B] minima over the (2*border+1)^2 shifts, cpsnr[B] maximum, arg_l1[B]/arg_l2[B] the
 * arg-min shift ids (i*(2*border+1)+j), mean_l1/mean_l2 the batch means (the two loss scalars).     */
int probav_shift_loss_forward(const float* hr, const uint8_t* mask, const float* pred, int batch, int size,
                              int border, int bit_depth, float* l1, float* l2, float* cpsnr, int32_t* arg_l1,
                              int32_t* arg_l2, float* mean_l1, float* mean_l2, void* stream);
/* gradient of mean_l1 (which=1) or mean_l2 (which=2) w.r.t. pred; `upstream` = device scalar or NULL */
int probav_shift_loss_backward(const float* hr, const uint8_t* mask, const float* pred, const int32_t* arg,
                               int batch, int size, int border, int which, const float* upstream,
                               float* dpred, void* stream);
/* cfg loss = sobel_l1_mix: Losses.shiftCompensatedL1EdgeLoss                     models/loss.py:86-97,126-137,214-219
 * per sample min over the (2*border+1)^2 shifts of  pi * L1 + (1 - pi) * sum|sobel_edges(HR) - sobel_edges(corrected SR)| / n
 * (tf.image.sobel_edges: REFLECT-padded 3x3 correlations; pi = Losses.pi = 0.7).  loss [batch], arg [batch],
 * mean: TWO floats (mean over the batch, scratch).  The backward differentiates the arg-min shift, bias term included.   */
int probav_shift_l1edge_forward(const float* hr, const uint8_t* mask, const float* pred, int batch, int size, int border,
                                float pi, float* loss, int32_t* arg, float* mean, void* stream);
int probav_shift_l1edge_backward(const float* hr, const uint8_t* mask, const float* pred, const int32_t* arg, int batch,
                                 int size, int border, float pi, const float* upstream, float* dpred, void* stream);
/* cfg loss = l1msssim: Losses.shiftCompensatedRevSSIM                             models/loss.py:99-124,189-212
 * ONE scalar for the batch: min over the shifts of  eta * (1 - sum_{scale,sample} luminance * prod_scale(contrast * structure) / B)
 * + (1 - eta) * weighted L1 / (B * (2^bit_depth - 1)), with the reference's five exponential windows (its quirks restated in
 * kernels_small.hip).  scratch: probav_revssim_scratch_bytes() bytes, written by the forward and read by the backward; loss and arg
 * are one element each (the batch shares the shift).                                                                               */
size_t probav_revssim_scratch_bytes(int batch, int border);
int probav_revssim_forward(const float* hr, const uint8_t* mask, const float* pred, int batch, int size, int border, int bit_depth,
                           float eta, void* scratch, size_t scratch_bytes, float* loss, int32_t* arg, void* stream);
int probav_revssim_backward(const float* hr, const uint8_t* mask, const float* pred, const int32_t* arg, const void* scratch,
                            int batch, int size, int border, int bit_depth, float eta, const float* upstream, float* dpred, void* stream);
/* replaces optimizer.apply_gradients with Keras Nadam                   models/trainClass.py:132, train.py:79-81
 * in place on the flat parameter buffer; m, v = first / second moment slots (n floats each).  The caller supplies the
 * step-dependent scalars of SURVEY.md A.5 (computed in double): c_g = (1-mu_t)/(1-Pi_t), c_m = mu_{t+1}/(1-Pi_t*mu_{t+1}),
 * c_v = 1/(1-beta2^t).  The update is  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  theta -= lr (c_g g + c_m m) / (sqrt(c_v v) + eps),
 * which also is Keras Adam (c_g = 0, c_m = sqrt(1-b2^t)/(1-b1^t), c_v = 1) and plain SGD (c_g = 1, c_m = 0, c_v = 0, eps = 1): the
 * three optimizers of train.py:77-83 are one fused launch.                                                           */
int probav_nadam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, float c_g, float c_m, float c_v, void* stream);
/* replaces tf.clip_by_value(sr, 0, 2**16); tf.round(sr)                 test.py:118-119             */
int probav_clip_round(const float* in, float* out, size_t n, float lo, float hi, void* stream);

/* ---- single operators (what the engine is made of; exported for parity tests) ------------------- */
/* geometry: int32[17] = N, Hi,Wi,Ti,Cin, Ho,Wo,To,Cout, kh,kw,kt, ph,pw,pt, reflect_hw, relu        */
/* y = act(conv(x * [gate>0], w) + bias) + skip; impl 0 = direct, 1 = MFMA row-tile, 2 = MFMA strip, 3 = x6 (strip or row-tile),
 * 4 = H3 (the same kernels with three products of scaled fp16 piece pairs; operand maxima are measured by the library) */
int probav_conv3d_forward(const int32_t geom[17], const float* x, const float* gate, const float* w,
                          const float* bias, const float* skip, float* y, int impl, void* stream);
size_t probav_conv3d_wgrad_scratch_bytes(const int32_t geom[17], int impl);
int probav_conv3d_wgrad(const int32_t geom[17], const float* x, const float* dy, const float* gate,
                        float* dw, float* db, void* scratch, size_t scratch_bytes, int impl, void* stream);
/* fused expConv_i (1x1x1, 32->256) + ReLU + decConv_i (1x1x1, 256->D<=26)      models/modelsTF.py:179-183
 * x [nvox,32], w1 [32,256], b1 [256], w2 [256,D], b2 [D] -> dec [nvox,D]; the 256-channel tensor never reaches HBM
 * impl 2 = fp32 MFMA, 3 = fp32 products as six bf16-piece products on the bf16 MFMA pipe ("x6", same accuracy class),
 * 4 = three products of scaled fp16 piece pairs ("H3", same accuracy class) */
/* vox_per_sample: voxels of one patch (nvox must be a multiple; 0 = treat the call as one sample): the unit the H3 arithmetic scales by */
int probav_pw_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* dec,
                      int64_t nvox, int64_t vox_per_sample, int D, int impl, void* stream);
/* its reverse pass: d_dec [nvox,D], d_skip [nvox,32] (gradient arriving over the residual connection)
 * -> dx = d_skip + dL/dx [nvox,32], dw1 [32,256], db1 [256], dw2 [256,D], db2 [D]
 * ACCURACY OF THE FILTER GRADIENTS, impl 4 (declared bound; tests/test_gpu_h3_range.py::test_pointwise_filter_gradients_slice_by_slice
 * asserts it).  dx, db1, db2 and every slice of dw1 / dw2 whose channel is within 2^-18 of its sample's largest value are at fp32 level
 * (<= 5e-6 of the slice's own maximum, the bar impl 2 and 3 meet on every slice).  dw1 (a ROW = one input channel of x) and dw2 (a
 * COLUMN = one channel of d_dec) contract over the voxels, but the kernel cuts x and d_dec into their two fp16 pieces ONCE per tile
 * with the per-SAMPLE power-of-two scale that the products contracting over those channels need; a channel that sits 2^-k below its
 * sample's maximum keeps both pieces normal only down to k = 18, and loses one bit of its second piece per binade below: the row /
 * column of such a channel is resolved to <= 1e-4 of its own maximum at k = 24 (measured 4.8e-5 / 9.0e-5) -- still far inside
 * north_star's 1e-3, and invisible in a whole-tensor norm.  The cure of the 3x3x3 backward-filter kernel (second pieces lifted by 2^11,
 * the cross products in an accumulator set of their own) needs 256 more accumulator registers than a wave has.  impl 2 / 3: no such floor. */
size_t probav_pw_backward_scratch_bytes(int D);
int probav_pw_backward(const float* x, const float* d_dec, const float* d_skip, const float* w1, const float* b1,
                       const float* w2, float* dx, float* dw1, float* db1, float* dw2, float* db2, void* scratch,
                       size_t scratch_bytes, int64_t nvox, int64_t vox_per_sample, int D, int impl, void* stream);
/* weight normalisation of every layer of the engine: params -> weff, weffT, inv_norm (ws-internal
 * layouts, exported for tests): sizes probav_weff_count() floats and probav_cout_total() floats      */
int64_t probav_weff_count(const probav_engine* e);
int64_t probav_cout_total(const probav_engine* e);
int probav_wn_forward(probav_engine* e, const float* params, float* weff, float* weffT, float* inv_norm, void* stream);
int probav_wn_backward(probav_engine* e, const float* params, const float* dweff, const float* inv_norm,
                       float* grads, void* stream);

/* What runs on the engine's side stream (see the conventions at the top): 0 = nothing, 1 = the slab sums and the low-frequency residual
 * path, 2 (default) = also the backward-filter kernels of the 3x3x3 layers, whose results only the weight-norm backward at the very end
 * reads: at the lowest stream priority they fill the tails of the caller's chain (-3 % per step) -- and share the chip with the kernels
 * they run beside, so a per-kernel timing (bench.py's roofline leg, a rocprofv3 kernel summary) is taken in mode 1.                        */
int probav_engine_side_stream(probav_engine* e, int mode);

/* ---- measurement aid -------------------------------------------------------------------------------------------------------------- */
/* Enqueues `launches` launches of nothing but dependent v_mfma_f32_32x32x16_f16 on every compute unit (one wave per SIMD, `iters` x 16
 * MFMAs per wave, operands from `seed`: 128 x 16 bytes of fp16 data).  Timed by the caller (events on `stream`), it gives the matrix rate
 * THIS device sustains under load -- the boxes of a pool differ -- as launches * 256 * 4 * iters * 16 * 32768 FLOP / time.
 * `sink` receives 256 * 256 floats.  bench.py reports it as `sustained_mfma_tflops` beside the step time.                                */
int probav_mfma_probe(const void* seed, float* sink, int iters, int launches, void* stream);
/* The same with the MFMA shape as a parameter (ABI 4): shape 0 = v_mfma_f32_32x32x16_f16 (iters x 16 MFMAs of 32 768 FLOP per wave), shape 1 =
 * v_mfma_f32_16x16x32_f16 (iters x 32 MFMAs of 16 384 FLOP: the same FLOP per wave, the same cycles per FLOP).  Where the chip lowers its clock
 * under matrix load the clock it holds depends on the shape (MI355X_MICROARCH.md, DVFS give-back, item 7): bench.py reports both rates. */
int probav_mfma_probe_shape(const void* seed, float* sink, int iters, int launches, int shape, void* stream);

/* ---- introspection of a training forward pass (parity tests; tf.keras would expose these as layer outputs) -------------------- */
/* where a saved activation lives inside the caller's workspace after probav_forward(training=1): offset and length in floats.
 * kind: ACT = input of residual block `index` (index num_res_blocks = output of the last block; ACT 0 = relu(mainConv1), models/modelsTF.py:58),
 * DEC = decConv_index output (:182-183), RED = relu(convReducer_{index+1}) (:159-160), RESID1 = relu(residConv1) (:47)                 */
#define PROBAV_VIEW_ACT 0
#define PROBAV_VIEW_DEC 1
#define PROBAV_VIEW_RED 2
#define PROBAV_VIEW_RESID1 3
int probav_workspace_view(const probav_engine* e, int batch, int training, int kind, int index, int64_t* offset_floats, int64_t* count);
/* the post-ReLU hidden tile relu(expConv_block(x)) [B*(P+s)^2*T][256] (models/modelsTF.py:179-180) exactly as the fused forward kernel
 * of the current kernel family (3 or 4) evaluates it -- that tensor never reaches memory otherwise.  Call after probav_forward(training=1)
 * with the same workspace; the ReLU gates of the reverse pass are the signs of these values.                                          */
int probav_debug_hidden(probav_engine* e, const float* params, const void* ws /* saved state: read only */, size_t ws_bytes, int batch, int block, float* hidden,
                        float* dec_scratch /* [voxels][dec channels] floats: the launch's regular output, discarded */,
                        const void* wcache /* the weight cache the forward pass ran from, or NULL */, void* stream);
/* ABI 7.  Which arrangement probav_debug_hidden evaluates the tile in (kernel family 4): 0 (default) the 32x32x16 one, whose order of additions is the one the
 * reverse pass recomputes the tile -- and decides its ReLU gates -- in; 1 the forward kernel's own (16x16x32).  The two sum the same piece products in different
 * orders; a pre-activation that is zero to rounding can be open in one and closed in the other (tests/test_gpu_parity.py bounds how many, and how large).  Process-wide. */
int probav_debug_hidden_from_forward_kernel(int on);

#ifdef __cplusplus
}
#endif
#endif
