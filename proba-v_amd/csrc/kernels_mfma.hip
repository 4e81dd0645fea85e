// MFMA kernels (placeholder until the fp32-MFMA implicit-GEMM kernels land): nothing supported yet,
// so the engine routes every layer through the generic direct kernels.
#include "probav_common.h"
namespace probav {
bool mfma_conv_supported(const ConvGeom&) { return false; }
int mfma_conv_forward(const ConvGeom&, const float*, const float*, const float*, const float*, const float*, float*, hipStream_t)
{
    set_error("mfma_conv_forward: not built", hipSuccess);
    return PROBAV_EINVAL;
}
}  // namespace probav
