// ubench_valu_peak.hip -- chip-wide sustained fp32 VALU FMA rate: v_fma_f32 vs v_pk_fma_f32 (VGPR and SGPR-pair weight operand), the ceiling of
// the VALU-only ESPCN kernel B.   hipcc --offload-arch=gfx950 -O3 -o ubench_valu_peak tools/ubench_valu_peak.hip
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void valu_kernel(int iters, const float* __restrict__ w, float* out) {
    f32x2 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = {0.0f, 0.0f};
    const float xv = threadIdx.x * 0.001f;
    const f32x2 x2 = {xv, xv};
    const f32x2 xq = {w[16 + (threadIdx.x & 3)], w[20 + (threadIdx.x & 3)]}; // two different values in one register pair
    const f32x2 wv = {w[threadIdx.x & 7], w[(threadIdx.x & 7) + 8]}; // per-lane (VGPR) weights
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2 ws = {w[2 * k], w[2 * k + 1]}; // uniform (SGPR pair) weights
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) { // 2 scalar FMAs
                    acc[i].x = fmaf(xv, wv.x, acc[i].x);
                    acc[i].y = fmaf(xv, wv.y, acc[i].y);
                } else if (MODE == 1) {
                    acc[i] = __builtin_elementwise_fma(x2, wv, acc[i]);
                } else if (MODE == 2) {
                    acc[i] = __builtin_elementwise_fma(x2, ws, acc[i]);
                } else { // MODE 3: the activation is ONE half of a register pair, broadcast with op_sel (what ESPCN kernel B issues)
                    const f32x2 xb = {(i & 1) ? xq.y : xq.x, (i & 1) ? xq.y : xq.x};
                    acc[i] = __builtin_elementwise_fma(xb, ws, acc[i]);
                }
            }
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t prop;
    (void) hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out, *w;
    (void) hipMalloc(&out, sizeof(float) * 256 * cus * 16);
    (void) hipMalloc(&w, sizeof(float) * 64);
    (void) hipMemset(w, 0, sizeof(float) * 64);
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 4; ++mode)
        for (int wps : {1, 2, 4}) {
            const int blocks = cus * wps;
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(valu_kernel<0>, dim3(blocks), dim3(256), 0, 0, iters, w, out);
                else if (mode == 1) hipLaunchKernelGGL(valu_kernel<1>, dim3(blocks), dim3(256), 0, 0, iters, w, out);
                else if (mode == 2) hipLaunchKernelGGL(valu_kernel<2>, dim3(blocks), dim3(256), 0, 0, iters, w, out);
                else hipLaunchKernelGGL(valu_kernel<3>, dim3(blocks), dim3(256), 0, 0, iters, w, out);
            };
            launch();
            (void) hipDeviceSynchronize();
            (void) hipEventRecord(e0);
            launch();
            (void) hipEventRecord(e1);
            (void) hipEventSynchronize(e1);
            float ms = 0;
            (void) hipEventElapsedTime(&ms, e0, e1);
            const double flops = static_cast<double>(blocks) * 256 * iters * 4.0 * 8 * 2 /*FMAs*/ * 2;
            printf("%-28s waves/SIMD=%d: %7.3f ms  %7.1f TFLOP/s\n", mode == 0 ? "2x v_fma_f32" : mode == 1 ? "v_pk_fma_f32 (VGPR weights)" : mode == 2 ? "v_pk_fma_f32 (SGPR weights)" : "v_pk_fma_f32 (SGPR w, op_sel x)", wps, ms,
                   flops / ms / 1e9);
        }
    return 0;
}
