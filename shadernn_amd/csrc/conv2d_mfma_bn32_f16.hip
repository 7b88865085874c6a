// conv2d_mfma_bn32_f16.hip -- the 32-output-channel-wide f16 variants of conv2d_mfma_kernel (WM=4, WN=1 waves, MT=1 x NT=1 register tiles per
// wave); the kernel template and the variant table are in conv2d_mfma_kernel.h, the plan builder in conv2d_mfma.hip.
#include "conv2d_mfma_kernel.h"

namespace snnhip {

mfma_detail::KernelFn pick_conv2d_mfma_bn32_f16(int c8, int r, bool simple, int taps) { return mfma_detail::pick_kernel<4, 1, 1, 1, true>(c8, r, simple, taps); }

} // namespace snnhip
