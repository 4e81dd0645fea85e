// Shared declarations of the gfx950 HIP library behind include/probav_hip.h.
// Layout conventions (all fp32):
//   activations  [N][H][W][T][C]  C innermost  (the reference's NDHWC with D,H,W == H,W,T)
//   kernels      [kh][kw][kt][Cin][Cout]       (Keras layout; models/modelsTF.py:191-197)
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <stdint.h>
#include <stddef.h>

#define PROBAV_OK 0
#define PROBAV_EINVAL (-1)      // bad argument (shape / null pointer / unsupported config)
#define PROBAV_ENOSPACE (-2)    // workspace too small
#define PROBAV_EHIP (-3)        // a HIP call failed; hipGetLastError() text via probav_last_error()

// Geometry of one stride-1 cross-correlation over (H, W, T).
// input coordinate = output coordinate + tap - pad ; outside [0, dim): zero, or mirrored on H/W when
// reflect_hw is set (tf.pad 'reflect', models/modelsTF.py:157-158 folded into the indexing).
struct ConvGeom {
    int N;
    int Hi, Wi, Ti, Cin;
    int Ho, Wo, To, Cout;
    int kh, kw, kt;
    int ph, pw, pt;
    int reflect_hw;
    int relu;
    int reflect_t;      // the depth pad mirrors too (only the experimental 19-frame reducer, models/modelsTF.py:76-121; direct kernels only)
};

namespace probav {

// amax slots of one launch (H3 arithmetic, x6_device.h): bit patterns of largest magnitudes (non-negative floats), kept in device memory.
//   x : the activation operand, ONE SLOT PER SAMPLE (x[n] = largest |value| of patch n)
//   w : forward / backward-data kernels: the filter, ONE SLOT PER OUTPUT COLUMN of the packed matrix (w[col]);
//       backward-filter kernel: the incoming gradient dY, one slot per sample like x
//   y : receives (atomicMax) the largest output magnitude of every sample (y[n]); written by any kernel when set
struct Amax { const unsigned* x = nullptr; const unsigned* w = nullptr; unsigned* y = nullptr; };
// largest |x[n][i]|, i < per_sample, -> slots[n] for n < N (atomicMax; the slots must have been zeroed), for tensors whose producer does not report it
int amax_tensor(const float* x, size_t per_sample, int N, unsigned* slots, hipStream_t s);
// largest |w[r][c]| over the rows of a row-major [rows][cols] matrix -> slots[c] (cols <= 256; plain store)
int amax_columns(const float* w, long rows, int cols, unsigned* slots, hipStream_t s);

void set_error(const char* what, hipError_t e);
// check_launch: hipGetLastError after a launch -> PROBAV_OK / PROBAV_EHIP.  It also reports (once set, sticky) a failed
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) of any kernel: note_attr_error() records it from inside the std::call_once
// blocks that raise the LDS limits, so the failure surfaces as what it is instead of as an opaque launch error later.
void note_attr_error(hipError_t e);
int check_launch(const char* what);
const char* last_error();

// ---- kernels_direct.hip : generic direct (VALU) convolution, any geometry ------------------------
// y = act( conv(x * [gate > 0], w) + bias ) + skip        (gate/bias/skip optional)
int conv3d_direct_forward(const ConvGeom& g, const float* x, const float* gate, const float* w,
                          const float* bias, const float* skip, float* y, hipStream_t s);
// dw[tap][ci][co] = sum_v x[v + tap][ci] * dy[v][co] * [gate[v][co] > 0] ; db[co] = sum_v dy*gate
// `partial` holds wgrad_partial_floats(g) floats of scratch.
size_t wgrad_partial_floats(const ConvGeom& g);
int conv3d_direct_wgrad(const ConvGeom& g, const float* x, const float* dy, const float* gate,
                        float* dw, float* db, float* partial, hipStream_t s);
int mfma_probe(const void* seed, float* sink, int iters, int launches, hipStream_t s, int shape = 0);      // shape 0: 32x32x16, 1: 16x16x32 (fp16)
// Slab reductions beside the main chain.  The backward-filter kernels leave one slab per workgroup; the few-microsecond kernels that sum
// them are needed only by the weight-norm backward at the very end, yet on the caller's stream each one sits between two chip-filling
// launches.  While a ReduceSide is active on the calling thread (the engine's backward pass), reduce_fork(s) records the point on s,
// makes the engine's side stream wait for it and returns the side stream; the reducing kernels are launched there and run in the
// gaps of the main chain.  reduce_join(s) makes s wait for everything forked.  Without an active context reduce_fork(s) is s.
// Every fork is an event record between two kernels of the caller's stream, and that costs the stream about 6.5 us of bubble (30 of them in a backward
// pass: 0.2 ms).  With `defer` set, reduce_later(s, fn) only QUEUES the launch fn(stream); reduce_flush(s) forks once and launches everything queued
// (each of those launches reads slabs that nobody writes again before the pass ends, so it may run any time after its producer); reduce_join flushes first.
struct ReduceSide { hipStream_t side; hipEvent_t ev[8]; hipEvent_t joined; int k; hipStream_t last; int defer; void* pending; };   // last: the stream of the latest SUCCESSFUL fork since the join (reduce_fork_adjacent); pending: the queued launches (kernels_small.hip)
void reduce_side_activate(ReduceSide* ctx);                 // nullptr deactivates
hipStream_t reduce_fork(hipStream_t s);
hipStream_t reduce_fork_adjacent(hipStream_t s);           // the same point as the caller's previous reduce_fork(s) (nothing enqueued on s in between): no new event
int reduce_later(hipStream_t s, std::function<int(hipStream_t)> fn);      // fn(reduce_fork(s)) now, or queued until the next reduce_flush (ReduceSide::defer)
int reduce_flush(hipStream_t s);                            // one fork for everything queued
// One slab sum: dst[i] = the fixed-order fp64 sum over c < slabs of src[c * stride + i], i < count (bitwise reproducible; any alignment).  slab_sum_later launches the
// jobs as ONE kernel on reduce_fork(s) -- or, while launches are being deferred, adds them to the batch that the next reduce_flush launches as one kernel ON THE LAUNCH
// STREAM (the backward pass's 27 slab sums of 15-22 MB each were 27 launches at 1.5 TB/s on the side stream; a flush's worth of them in one launch streams at 3.7 TB/s,
// and on the launch stream it costs no event and nothing is left for the join to wait for: docs/notebook_r1-r5.md section 4.00, the slab sums)
struct SlabSumJob { const float* src; float* dst; long stride; int count; int slabs; };
int slab_sum_later(hipStream_t s, const SlabSumJob* jobs, int njobs);
void reduce_free_pending(ReduceSide* ctx);                // the engine is going away: release the queue object
void reduce_drop_pending();                                 // an aborted pass: forget the queue (its closures may point into a dead frame)
int reduce_join(hipStream_t s);
// mainConv1 forward (one input channel -> 32, 3x3x3, zero pads of 1): dedicated store-bound kernel; amax = per-sample slots of y or null
bool conv3d_cin1_forward_supported(const ConvGeom& g);
int conv3d_cin1_forward(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, unsigned* amax, hipStream_t s);
// the low-frequency residual path (models/modelsTF.py:45-53) as ONE launch each way (kernels_direct.hip): mn [N][Hin][Hin][Cx] -> r1 [N][Hin-2]^2[9] (ReLU) -> r2 -> r3;
// weights [3][3][1][Cin][9] (the effective, weight-normalised filters).  Backward: d r3 = dtail -> the six gradients; slabs: resid_path_slab_floats(N, Cx) floats.
bool resid_path_supported(int Hin, int Cx, int C);
size_t resid_path_slab_floats(int N, int Cx);
int resid_path_forward(int N, int Hin, int Cx, const float* mn, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                       float* r1, float* r2, float* r3, hipStream_t s);
int resid_path_backward(int N, int Hin, int Cx, const float* mn, const float* r1, const float* r2, const float* dtail, const float* w2, const float* w3,
                        float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, float* slabs, hipStream_t s);
// geometries for which conv3d_direct_wgrad runs a dedicated kernel that beats the matrix kernels (one input channel: mainConv1)
// upscaleConv1 (32 -> 9, valid, depth 3 -> 1) and its backward-data (9 -> 32, full, depth 1 -> 3; w = the flipped, channel-swapped filter): small VALU kernels
bool conv3d_up_forward_supported(const ConvGeom& g);
int conv3d_up_forward(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, hipStream_t s);
bool conv3d_up_bwd_data_supported(const ConvGeom& g);
int conv3d_up_bwd_data(const ConvGeom& g, const float* dy, const float* wT, const float* bias, float* dx, unsigned* amax, hipStream_t s);
bool conv3d_direct_wgrad_is_tuned(const ConvGeom& g);

// ---- kernels_small.hip ---------------------------------------------------------------------------
struct WnLayer {            // one weight-normalised layer inside the flat parameter buffer
    int g_off, v_off, b_off;    // offsets into params / grads (floats)
    int w_off;                  // offset into weff / weffT / dweff (floats) == running sum of K*Cout
    int n_off;                  // offset into inv_norm (floats) == running sum of Cout
    int K;                      // taps * Cin
    int Cin, Cout, taps;
    int r_off;                  // offset into the per-input-channel amax array == running sum of Cin
};
// amax (optional): zeroed slots [2 nlayers + cout_total + cin_total]; receives per layer l the largest |effective weight| (slot l) and
// |bias| (slot nlayers + l), per output channel the largest |weight| of its column (slot 2 nlayers + n_off + co) and per input
// channel that of its rows (slot 2 nlayers + cout_total + r_off + ci)
int wn_forward(const WnLayer* d_layers, int nlayers, int max_cout_total, int cin_total, const float* params,
               float* weff, float* weffT, float* inv_norm, unsigned* amax, hipStream_t s);
// the optimizer update of EVERY parameter (rule and coefficients of nadam_step) fused with the weight normalisation of the updated
// parameters: one wave per (layer, output channel) updates its g, bias and filter column, then writes the column of weff / weffT /
// inv_norm and the amax slots exactly as wn_forward does
int optimizer_wn_step(const WnLayer* d_layers, int nlayers, int cout_total, int cin_total, float* params, const float* grad, float* m, float* v,
                      float lr, float b1, float b2, float eps, float c_g, float c_m, float c_v,
                      float* weff, float* weffT, float* inv_norm, unsigned* amax, hipStream_t s);
int wn_backward(const WnLayer* d_layers, int nlayers, int max_cout_total, const float* params,
                const float* dweff, const float* inv_norm, float* grads, hipStream_t s);
int head_forward(const float* x, float* xn, float* mn, int nvox_hw, int T, int C, float mean, float stdv, hipStream_t s, unsigned* zero = nullptr, int nzero = 0);   // zero[0 .. nzero): cleared in the same launch (the pass's per-sample amax slots)
int tail_forward(const float* up, const float* r3, float* y, int N, int P, int scale, float mean, float stdv, hipStream_t s);
int tail_backward(const float* dy, float* dtail, int N, int P, int scale, float stdv, hipStream_t s, unsigned* zero = nullptr, int nzero = 0);
int reflect_fold(const float* dpad, float* dx, int N, int H, int W, int TC, unsigned* amax /* optional: N slots, receive max |dx| per sample */, hipStream_t s);
// general form: gradient of tf.pad(x, [ph, pw, pt] 'reflect') folded back onto x [N,H,W,T,C]; pads <= 2
int reflect_fold3(const float* dpad, float* dx, int N, int H, int W, int T, int C, int ph, int pw, int pt, hipStream_t s);
int clip_round(const float* in, float* out, size_t n, float lo, float hi, hipStream_t s);
int shift_loss_forward(const float* hr, const uint8_t* mask, const float* pred, int B, int S, int border,
                       float* l1_per_sample, float* l2_per_sample, float* cpsnr_per_sample,
                       int* arg_l1, int* arg_l2, float* mean_l1, float* mean_l2, float max_val, hipStream_t s);
int shift_loss_backward(const float* hr, const uint8_t* mask, const float* pred, const int* arg, int B, int S,
                        int border, int which /*1 = L1, 2 = L2*/, const float* upstream /*device scalar or null*/,
                        float* dpred, hipStream_t s);

int shift_l1edge_forward(const float* hr, const uint8_t* mask, const float* pred, int B, int S, int border, float pi,
                         float* loss, int* arg, float* mean, float* scratch_mean2, hipStream_t s);
int shift_l1edge_backward(const float* hr, const uint8_t* mask, const float* pred, const int* arg, int B, int S, int border, float pi,
                          const float* upstream, float* dpred, hipStream_t s);

size_t revssim_scratch_bytes(int B, int border);
int revssim_forward(const float* hr, const uint8_t* mask, const float* pred, int B, int S, int border, float max_val, float eta,
                    double* scratch, float* loss, int* arg, hipStream_t s);
int revssim_backward(const float* hr, const uint8_t* mask, const float* pred, const int* arg, const double* scratch, int B, int S,
                     int border, float max_val, float eta, const float* upstream, float* dpred, hipStream_t s);

int nadam_step(float* theta, const float* grad, float* m, float* v, long n, float lr, float b1, float b2, float eps,
               float c_g, float c_m, float c_v, hipStream_t s);

}  // namespace probav
