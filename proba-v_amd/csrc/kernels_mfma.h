// Host-visible interface of kernels_mfma.hip (fp32-MFMA kernels + weight fragment packing).
#pragma once
#include "probav_common.h"

namespace probav {

enum { PACK_CONV = 0, PACK_PW_A_KCIN = 1, PACK_PW_A_KHCH = 2, PACK_PW_A_KOUT = 3, PACK_PW_A_CIN_KHCH = 4,
       // pre-split bf16 operand fragments of the x6 kernels (three truncation pieces per value, 16 B per lane and fragment)
       PACK_X6_PW_W1 = 10, PACK_X6_PW_W2 = 11, PACK_X6_CONV = 12, PACK_X6_PW_W2K = 13, PACK_X6_PW_W1C = 14, PACK_X6_CONVK = 15, PACK_X6_CONVP = 16,
       // the same fragment orders with two scaled fp16 pieces per value (H3 arithmetic, x6_device.h): type = PACK_X6_* + 10
       PACK_H3_PW_W1 = 20, PACK_H3_PW_W2 = 21, PACK_H3_CONV = 22, PACK_H3_PW_W2K = 23, PACK_H3_PW_W1C = 24, PACK_H3_CONVK = 25,
       // Cin = 25, K of a (dh, dw) group as ten 8-channel chunks: (dt, channels 8c .. 8c+7) for dt, c < 3, then [ch 24 of dt = 0, 1, 2 | 0 x 5]
       // (the operand order of the piece-ring strip kernel); same size as the K-concatenated form
       PACK_H3_CONVP = 26 };
constexpr long X6_PW_FRAG_WORDS = 8 * 2 * 3 * 64 * 4;      // [8 chunks][2 k-blocks][3 pieces][64 lanes] x 16 B
constexpr long X6_CONV_FRAG_WORDS = 27 * 2 * 3 * 64 * 4;   // [27 taps][2 k-blocks][3 pieces][64 lanes] x 16 B
constexpr long X6_CONVK_FRAG_WORDS = 9 * 5 * 3 * 64 * 4;   // Cin = 25, K = (dt, ci) concatenated: [9 (dh,dw)][5 k-blocks][3 pieces][64 lanes] x 16 B
constexpr long H3_PW_FRAG_WORDS = 8 * 2 * 2 * 64 * 4;      // two pieces instead of three
constexpr long H3_CONV_FRAG_WORDS = 27 * 2 * 2 * 64 * 4;
constexpr long H3_CONVK_FRAG_WORDS = 9 * 5 * 2 * 64 * 4;

// One packing job: effective weights (weff / weffT, layout [tap][Cin][Cout]) -> MFMA operand fragments.
struct PackJob {
    int type, src_is_T;
    long src_off, dst_off, count;     // offsets in floats; count = floats written (multiple of 256)
    int Cin, Cout, CC, KS, taps;
    int amax_slot;                    // PACK_H3_*: index of the source tensor's amax in the array handed to mfma_pack ...
    int amax_percol, ncol;            // ... or, when amax_percol, of the FIRST of ncol per-column slots: the lane that packs output column / row `col`
                                      //     of the matrix (conv: cout of the packed matrix; PW_W2: out d; PW_W1C: cin f) scales by slot amax_slot + col
};

// amax: largest magnitudes (float bit patterns) of the weights, per tensor and per column (wn_forward's layout), read by the PACK_H3_* jobs (may be null without such jobs)
int mfma_pack(const PackJob* d_jobs, int njobs, const float* weff, const float* weffT, float* wpack, const unsigned* amax, hipStream_t s);

bool mfma_conv_supported(const ConvGeom& g);
size_t mfma_conv_wfrag_floats(int Cin, int Cout);          // 0 = this channel configuration is not packed
void mfma_conv_pack_job(PackJob& J, int Cin, int Cout);    // fills type/Cin/Cout/CC/KS/taps/count
int mfma_conv_forward(const ConvGeom& g, const float* x, const float* gate, const float* wfrag, const float* bias,
                      const float* skip, float* y, const Amax& am, hipStream_t s);

// strip form (ring of input rows, flattened tiles, K split over wave pairs); same fragment layout as mfma_conv_forward
bool mfma_conv_strip_supported(const ConvGeom& g);
int mfma_conv_strip_forward(const ConvGeom& g, const float* x, const float* gate, const float* wfrag, const float* bias,
                            const float* skip, float* y, const Amax& am, hipStream_t s);
// row-tile kernel with a split-operand tap loop (32-channel inputs: reducers, upscale; any pads / reflect).  arith 1: X6,
// wfrag6 = PACK_X6_CONV / _CONVK fragments; arith 2: H3, PACK_H3_* fragments and am.x / am.w set (x6_device.h)
bool x6_conv_rowtile_supported(const ConvGeom& g);
int x6_conv_rowtile_forward(const ConvGeom& g, const float* x, const float* gate, const float* wfrag6, const float* bias,
                            const float* skip, float* y, int arith, const Amax& am, hipStream_t s);
// the strip kernel with a split-operand tap loop (same fragments; when x6_strip_wants_tap_fragments(g, arith) the filters of a
// 25-channel layer must be the per-tap PACK_H3_CONV form, not PACK_H3_CONVK: the H3 piece-ring kernel serves the call)
bool x6_strip_wants_tap_fragments(const ConvGeom& g, int arith);
int x6_conv_strip_forward(const ConvGeom& g, const float* x, const float* gate, const float* wfrag6, const float* bias,
                          const float* skip, float* y, int arith, const Amax& am, hipStream_t s);

bool mfma_wgrad_supported(const ConvGeom& g);
size_t mfma_wgrad_partial_floats(const ConvGeom& g);
int mfma_conv_wgrad(const ConvGeom& g, const float* x, const float* dy, const float* gate, float* dw, float* db,
                    float* partial, hipStream_t s);

bool mfma_pw_supported(int F, int E, int D);
int mfma_pw_forward(const float* x, const float* w1frag, const float* w2frag, const float* b1, const float* b2, float* dec,
                    long nvox, int D, hipStream_t s);

size_t mfma_pw_backward_slab_floats(int D);
// dX = dOut + d(expand,decay)/dX ; dW1 [32][256], dW2 [256][D], db1 [256], db2 [D] (all overwritten)
int mfma_pw_backward(const float* x, const float* dT, const float* dOut, const float* w1kcin, const float* w2kout,
                     const float* w1khch, const float* b1, float* dX, float* dW1, float* dW2, float* db1, float* db2,
                     float* slabs, long nvox, int D, hipStream_t s);

// fixed-order fp64 sum of `slabs` backward-filter slabs ([nw] each at partial, [Cout] each at partial_b)
int mfma_wgrad_reduce(const float* partial, const float* partial_b, float* dw, float* db, long nw, int Cout, int slabs, hipStream_t s);
// fixed-order fp64 sum of the per-workgroup slabs of the fused pointwise backward (both the fp32 and the x6 kernel write them)
int mfma_pw_backward_reduce(const float* slabs, int D, float* dW1, float* dW2, float* db1, float* db2, hipStream_t s);
int mfma_pw_backward_grid();

// 3x3x3 'same' convolution of the residual blocks (25 -> 32 channels, and its backward-data 32 -> 25 / 32) as ONE-WAVE-PER-SIMD kernel (kernels_cw4.hip; H3
// arithmetic only): the filter's first pieces stay in registers for the whole launch.  x6_conv_strip_forward dispatches to it when cw4_conv_supported();
// cw4_set_enabled(0) (or PROBAV_GEN1=1 in the environment, which also selects the general pointwise backward) keeps conv3_pp_kernel.
// wfrag: PACK_H3_CONVP (25 input channels) / PACK_H3_CONV (32) fragments, as conv3_pp_kernel reads them
bool cw4_conv_supported(const ConvGeom& g, const float* gate);
bool cw4_enabled();
void cw4_set_enabled(int on);
int cw4_conv_forward(const ConvGeom& g, const float* x, const float* wfrag, const float* bias, const float* skip, float* y, const Amax& am, hipStream_t s);

}  // namespace probav
