// Host-visible interface of kernels_x6.hip: fp32 products evaluated as six bf16 piece products on the bf16 MFMA pipe.
#pragma once
#include "probav_common.h"
#include "kernels_mfma.h"      // PACK_X6_* fragment packing lives in mfma_pack()

namespace probav {

int x6_pw_forward(const float* x, const float* w1frag, const float* w2frag, const float* b1, const float* b2, float* dec,
                  long nvox, int D, hipStream_t s);

}  // namespace probav
