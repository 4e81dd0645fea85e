// Host-visible interface of kernels_x6.hip: fp32 products evaluated as six bf16 piece products on the bf16 MFMA pipe.
#pragma once
#include "probav_common.h"

namespace probav {

enum { X6_PW_W1 = 0, X6_PW_W2 = 1 };

// One packing job: effective fp32 weights -> pre-split bf16 A-operand fragments (offsets/count in 4-byte words).
struct X6PackJob {
    int type, src_is_T;
    long src_off, dst_off, count;
    int Cin, Cout;
};
constexpr long X6_PW_FRAG_WORDS = 8 * 2 * 3 * 64 * 4;      // [8 chunks][2 k-blocks][3 pieces][64 lanes] x 16 B

int x6_pack(const X6PackJob* d_jobs, int njobs, const float* weff, const float* weffT, float* wpack, hipStream_t s);

int x6_pw_forward(const float* x, const float* w1frag, const float* w2frag, const float* b1, const float* b2, float* dec,
                  long nvox, int D, hipStream_t s);

}  // namespace probav
