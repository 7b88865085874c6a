// ubench_mfma_f16.hip -- developer micro-benchmark (round 6): cycles per MFMA per wave on one SIMD for the f16 forms a split-precision
// (x = hi + lo, three f16 products, fp32 accumulate) pointwise stage could use -- v_mfma_f32_16x16x16_f16 (the CDNA3 form) and
// v_mfma_f32_16x16x32_f16 (gfx950) -- beside v_mfma_f32_16x16x4_f32, each with K independent packed fp32 FMAs issued per MFMA
// (does fp32 VALU work hide beside an f16 MFMA stream, which it does not beside an fp32 one: tools/ubench_issue.hip).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_f16.hip -o build/ubench_mfma_f16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int K, int KIND>
__global__ __launch_bounds__(256) void kern(float* out, long long* cyc, int iters) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f16x4 a4, b4;
    f16x8 a8, b8;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a4[i] = static_cast<_Float16>(a + i); b4[i] = static_cast<_Float16>(b - i); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { a8[i] = static_cast<_Float16>(a + i); b8[i] = static_cast<_Float16>(b - i); }
    f32x2 pk[8], pb = {b, a}, pc = {a, b};
#pragma unroll
    for (int i = 0; i < 8; ++i) pk[i] = f32x2{a + i, b + i};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int mm = 0; mm < 32; ++mm) { // (32 MFMAs per trip: a 4-MFMA loop body measures the loop, not the pipe)
            const int m = mm & 3;
            if (KIND == 0) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            if (KIND == 1) acc[m] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[m], 0, 0, 0);
            if (KIND == 2) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[m], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[(m * K + k) & 7]) : "v"(pb), "v"(pc));
        }
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += pk[i].x + pk[i].y;
#pragma unroll
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int KIND>
void run(float* out, long long* cyc, int blocksPerCU) {
    const int iters = 250, nb = 256 * blocksPerCU;
    hipLaunchKernelGGL((kern<K, KIND>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((kern<K, KIND>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    static long long h[4096];
    hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < nb; ++i) s += h[i];
    printf("%-12s + %2d v_pk_fma_f32 per MFMA, %d wave(s)/SIMD: %.1f cycles per MFMA per wave\n", KIND == 0 ? "16x16x4 f32" : KIND == 1 ? "16x16x16 f16" : "16x16x32 f16", K,
           blocksPerCU, s / nb / iters / 32);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&cyc, 4096 * 8);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>(out, cyc, w);
        run<0, 1>(out, cyc, w);
        run<0, 2>(out, cyc, w);
        run<1, 0>(out, cyc, w);
        run<1, 1>(out, cyc, w);
        run<1, 2>(out, cyc, w);
        run<2, 1>(out, cyc, w);
        run<2, 2>(out, cyc, w);
        run<4, 0>(out, cyc, w);
        run<4, 1>(out, cyc, w);
        run<4, 2>(out, cyc, w);
        run<8, 2>(out, cyc, w);
    }
    return 0;
}
