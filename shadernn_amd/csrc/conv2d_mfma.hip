// conv2d_mfma.hip -- placeholder until the fp32-MFMA implicit-GEMM kernels land.
#include "snnhip_internal.h"
namespace snnhip {
int make_conv2d_mfma_plan(snnhip_ctx*, const ConvGeom&, const float*, const std::vector<float>&, snnhip_plan**) { return SNNHIP_E_UNSUPPORTED; }
} // namespace snnhip
