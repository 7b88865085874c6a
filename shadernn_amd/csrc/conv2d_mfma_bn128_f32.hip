// conv2d_mfma_bn128_f32.hip -- the 128-output-channel-wide f32 variants of conv2d_mfma_kernel (WM=2, WN=2 waves, MT=2 x NT=2 register tiles per
// wave); the kernel template and the variant table are in conv2d_mfma_kernel.h, the plan builder in conv2d_mfma.hip.
#include "conv2d_mfma_kernel.h"

namespace snnhip {

mfma_detail::KernelFn pick_conv2d_mfma_bn128_f32(int c8, int r, bool simple, int taps) { return mfma_detail::pick_kernel<2, 2, 2, 2, false>(c8, r, simple, taps); }

} // namespace snnhip
