// Host-visible interface of kernels_x6.hip: fp32 products evaluated as six bf16 piece products on the bf16 MFMA pipe.
#pragma once
#include "probav_common.h"
#include "kernels_mfma.h"      // PACK_X6_* fragment packing lives in mfma_pack()

namespace probav {

// amax slots of the fused pointwise pair (H3 arithmetic, x6_device.h): x = the 32-channel input, ONE SLOT PER SAMPLE; dt = the incoming
// gradient (backward), one slot per sample; w1, w2, b1 = expConv weights, decConv weights, expConv bias, one slot per TENSOR;
// w2c = decConv weights per output column d (forward: PACK_H3_PW_W2 is cut per column); w1r = expConv weights per input row f (backward:
// PACK_H3_PW_W1C is cut per row); y = slots receiving the output's amax per sample (optional)
struct PwAmax { const unsigned* x = nullptr; const unsigned* w1 = nullptr; const unsigned* w2 = nullptr; const unsigned* b1 = nullptr;
                const unsigned* dt = nullptr; const unsigned* w2c = nullptr; const unsigned* w1r = nullptr; unsigned* y = nullptr; };

// arith 1: X6 (PACK_X6_PW_* fragments), 2: H3 (PACK_H3_PW_* fragments, am filled in).  vps = voxels per sample (nvox % vps == 0; 0 = one
// sample): tiles never straddle samples.  hdump (optional, tests): receives the post-ReLU hidden tile [nvox][256]
int x6_pw_forward(const float* x, const float* w1frag, const float* w2frag, const float* b1, const float* b2, float* dec,
                  long nvox, long vps, int D, int arith, const PwAmax& am, hipStream_t s, float* hdump = nullptr);

// tests: the hidden-tile dump of x6_pw_forward(…, hdump) (H3) comes from the forward kernel itself (pw_fwd_h3k_kernel, 16x16x32) instead of the 32x32x16 arrangement
void x6_pw_dump_from_forward_kernel(int on);

// reverse pass of the fused pair; w1f = PACK_X6_PW_W1, w2kf = PACK_X6_PW_W2K, w1cf = PACK_X6_PW_W1C fragments;
// slabs: mfma_pw_backward_slab_floats(D) floats
int x6_pw_backward(const float* x, const float* dT, const float* dOut, const float* w1f, const float* w2kf, const float* w1cf,
                   const float* b1, float* dX, float* dW1, float* dW2, float* db1, float* db2, float* slabs, long nvox, long vps, int D,
                   int arith, const PwAmax& am, hipStream_t s);

// the same reverse pass as ONE-WAVE-PER-SIMD kernel (kernels_pw4.hip; H3 arithmetic only): four independent waves per workgroup, each with a run of
// tiles of one sample and all eight hidden chunks, dW1 / dW2 of the run in 256 accumulator registers.  x6_pw_backward dispatches to it when
// pw4_backward_supported(); pw4_set_enabled(0) (or PROBAV_GEN1=1 in the environment) keeps the general eight-wave form pw_bwd_x6_kernel<H3>
bool pw4_backward_supported(long nvox, long vps, int D);
bool pw4_enabled();
void pw4_set_enabled(int on);
int pw4_backward(const float* x, const float* dT, const float* dOut, const float* w1f, const float* w2kf, const float* w1cf,
                 const float* b1, float* dX, float* dW1, float* dW2, float* db1, float* db2, float* slabs, long nvox, long vps, int D,
                 const PwAmax& am, hipStream_t s);

// the fused pointwise forward as ONE-WAVE-PER-SIMD kernel (kernels_pf4.hip; H3 arithmetic only): the arithmetic of pw_fwd_x6_kernel<H3> -- the order in which the reverse
// pass recomputes the hidden tile --, the weight fragments in registers, no LDS traffic in the tile loop.  x6_pw_forward dispatches to it when pf4_forward_supported();
// pf4_set_enabled(0) (or PROBAV_GEN1=1 / pw in the environment) keeps pw_fwd_h3k_kernel
bool pf4_forward_supported(long nvox, long vps, int D);
bool pf4_enabled();
void pf4_set_enabled(int on);
int pf4_forward(const float* x, const float* w1frag, const float* w2frag, const float* b1, const float* b2, float* dec, long nvox, long vps, int D,
                const PwAmax& am, hipStream_t s, float* hdump = nullptr);

// backward-filter of a 3x3x3 convolution with Cin = 25 or 32 and Cout = 32 (normConv, reducers; pads 0/1, reflect, ReLU gate);
// partial: x6_wgrad_partial_floats(g) floats
bool x6_wgrad_supported(const ConvGeom& g);
size_t x6_wgrad_partial_floats(const ConvGeom& g);
// arith 1: X6; 2: H3 with am.x = per-sample amax slots of x, am.w = per-sample amax slots of dY (g.N of each)
int x6_conv_wgrad(const ConvGeom& g, const float* x, const float* dy, const float* gate, float* dw, float* db, float* partial,
                  int arith, const Amax& am, hipStream_t s);

// the backward-filter of the residual blocks (25 -> 32 channels, 'same' padding, depth 9 or 7) and of the reducers (32 -> 32, tf.pad(REFLECT) rows / columns, no depth pads,
// output depth 7 / 5 / 3, dY masked by `gate` = the layer's output) on rows of 22 columns and at most 256 samples as ONE-WAVE-PER-SIMD kernel
// (kernels_wg4.hip; H3 arithmetic only): a conflict-free piece image of the input ring, dY straight from memory, one instruction stream per output row.
// x6_conv_wgrad dispatches to it when wg4_wgrad_supported(); wg4_set_enabled(0) (or PROBAV_GEN1=1 / wg in the environment) keeps conv3_wgrad_x6_kernel.
// partial: x6_wgrad_partial_floats(g) floats, as for the general form
bool wg4_wgrad_supported(const ConvGeom& g, const float* gate);
bool wg4_enabled();
void wg4_set_enabled(int on);
int wg4_conv_wgrad(const ConvGeom& g, const float* x, const float* dy, const float* gate, float* dw, float* db, float* partial, const Amax& am, hipStream_t s);

}  // namespace probav
