// ubench_mfma_peak.hip -- chip-wide sustained MFMA rate (no memory traffic): the practical ceiling that the conv kernels' roofline fractions
// should be read against (power / clock management keeps a fully busy MI355X below its nominal 2.4 GHz x 256 CU peak).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_mfma_peak tools/ubench_mfma_peak.hip && ./ubench_mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void peak_kernel(int iters, float* out) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = static_cast<_Float16>(threadIdx.x * 0.001f + j);
        b[j] = static_cast<_Float16>(threadIdx.x * 0.002f - j);
    }
    const float af = threadIdx.x * 0.001f, bf = threadIdx.x * 0.002f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[i], 0, 0, 0);
            }
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int wavesPerSimd = 1; wavesPerSimd <= 2; ++wavesPerSimd)
            for (int iters : {2000, 20000, 100000}) {
                const int blocks = cus * wavesPerSimd;
                auto launch = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(peak_kernel<0>, dim3(blocks), dim3(256), 0, 0, iters, out);
                    else hipLaunchKernelGGL(peak_kernel<1>, dim3(blocks), dim3(256), 0, 0, iters, out);
                };
                launch();
                hipDeviceSynchronize();
                hipEventRecord(e0);
                launch();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const double flopsPer = mode == 0 ? 32.0 * 32 * 16 * 2 : 32.0 * 32 * 2 * 2;
                const double flops = static_cast<double>(blocks) * 4 /*waves*/ * iters * 16.0 * flopsPer;
                printf("%s waves/SIMD=%d iters=%6d: %8.3f ms  %8.1f TFLOP/s\n", mode == 0 ? "f16 32x32x16" : "f32 32x32x2 ", wavesPerSimd, iters, ms, flops / ms / 1e9);
            }
    return 0;
}
