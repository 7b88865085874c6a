// ubench_issue.hip -- developer micro-benchmark: how many independent VALU ops issue "for free" next to a stream of
// v_mfma_f32_16x16x4_f32 (4 accumulator chains), with 1 or 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int KIND>
__global__ __launch_bounds__(256) void kern(float* out, long long* cyc, int iters) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pk[8], pb = {b, a};
#pragma unroll
    for (int i = 0; i < 8; ++i) pk[i] = f32x2{a + i, b + i};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (KIND == 0) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            if (KIND == 2 || KIND == 3) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            if (KIND == 1) acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int r = (m * K + k) & 15;
                if (KIND < 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(b), "v"(a));
                if (KIND == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[r]) : "v"(b));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(pk[r & 7]) : "v"(pb));
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += pk[i].x + pk[i].y;
#pragma unroll
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int KIND = 0>
void run(float* out, long long* cyc, int blocksPerCU) {
    const int iters = 2000, nb = 256 * blocksPerCU;
    hipLaunchKernelGGL((kern<K, KIND>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((kern<K, KIND>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[4096];
    hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < nb; ++i) s += h[i];
    printf("%s K=%2d VALU per MFMA, %d wave(s)/SIMD: %.1f cycles per MFMA per wave\n", KIND == 1 ? "4x4x1  " : KIND == 2 ? "16x16x4+v_sub" : KIND == 3 ? "16x16x4+v_pk_add(neg)" : "16x16x4", K, blocksPerCU, s / nb / iters / 4);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&cyc, 4096 * 8);
    for (int w = 1; w <= 3; ++w) {
        run<8, 2>(out, cyc, w);
        run<8, 3>(out, cyc, w);
        run<16, 2>(out, cyc, w);
        run<16, 3>(out, cyc, w);
        run<0, 1>(out, cyc, w);
        run<2, 1>(out, cyc, w);
        run<4, 1>(out, cyc, w);
        run<8, 1>(out, cyc, w);
        run<0>(out, cyc, w);
        run<2>(out, cyc, w);
        run<4>(out, cyc, w);
        run<6>(out, cyc, w);
        run<8>(out, cyc, w);
        run<12>(out, cyc, w);
        run<16>(out, cyc, w);
    }
    return 0;
}
