// ubench_wide_core.hip -- can ONE wave per SIMD keep the fp16 matrix pipe full when both operands of a 4 x 2 register block come from LDS?
// (The question behind a loader-wave form of conv2d_wide_f16: compute waves that only ds_read and multiply.)  Per K step a wave reads 4 pixel
// operands + 2 weight operands (6 ds_read_b128, 6 KB) for 8 v_mfma_f32_32x32x16_f16; MODE 0 = no reads, 1 = reads of step s + 1 issued under the
// MFMAs of step s, 2 = the same with sched_group_barrier interleaving (one read per MFMA gap).  WPS waves per SIMD (blocks of 256 threads).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_wide_core tools/ubench_wide_core.hip && ./ubench_wide_core
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void core_kernel(int steps, int ldsFloat4, float* out) {
    extern __shared__ float4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < ldsFloat4; i += 256) lds[i] = make_float4(i * 1e-4f, 1.0f, 0.5f, 0.25f);
    __syncthreads();
    f32x16 acc[4][2];
    for (int t = 0; t < 4; ++t)
        for (int u = 0; u < 2; ++u)
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;
    const float4* ap = lds + wave * 832 + lane;            // pixel operands: 4 tiles, 64 float4 apart, four step slots of 192
    const float4* bp = lds + 4 * 832 + wave * 288 + lane;  // weight operands
    float4 a[4], b[2];
    for (int t = 0; t < 4; ++t) a[t] = ap[t * 64];
    for (int u = 0; u < 2; ++u) b[u] = bp[u * 64];
    for (int s = 0; s < steps; ++s) {
        const int o = ((s + 1) & 3) * 192; // four different step slots
        float4 an[4], bn[2];
        if (MODE >= 1) {
            for (int t = 0; t < 4; ++t) an[t] = ap[o + t * 64];
            for (int u = 0; u < 2; ++u) bn[u] = bp[(o >> 2) + u * 64];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&b[u]), *reinterpret_cast<const h8*>(&a[t]), acc[t][u], 0, 0, 0);
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        if (MODE >= 1) {
            for (int t = 0; t < 4; ++t) a[t] = an[t];
            for (int u = 0; u < 2; ++u) b[u] = bn[u];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float sres = 0.0f;
    for (int t = 0; t < 4; ++t)
        for (int u = 0; u < 2; ++u)
            for (int r = 0; r < 16; ++r) sres += acc[t][u][r];
    out[blockIdx.x * 256 + tid] = sres;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * cus * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int steps = 20000;
    for (int wps = 1; wps <= 2; ++wps)
        for (int mode = 0; mode < 3; ++mode) {
            const size_t ldsBytes = wps == 1 ? 100 * 1024 : 72 * 1024; // (the kernel uses 70 KB; 100 KB keeps a CU to one block)
            const int lf4 = 4 * 832 + 4 * 288;
            auto fn = mode == 0 ? core_kernel<0> : mode == 1 ? core_kernel<1> : core_kernel<2>;
            if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsBytes)); e != hipSuccess)
                printf("hipFuncSetAttribute(%zu): %s\n", ldsBytes, hipGetErrorString(e));
            auto launch = [&]() { hipLaunchKernelGGL(fn, dim3(cus * wps), dim3(256), ldsBytes, 0, steps, lf4, out); };
            launch();
            if (hipError_t e = hipDeviceSynchronize(); e != hipSuccess || hipGetLastError() != hipSuccess) printf("launch failed: %s\n", hipGetErrorString(e));
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = static_cast<double>(cus) * wps * 4 * steps * 8.0 * 32 * 32 * 16 * 2;
            printf("waves/SIMD=%d mode=%d (%s): %8.3f ms  %8.1f TFLOP/s  (%.1f cycles per MFMA at 2.4 GHz)\n", wps, mode,
                   mode == 0 ? "no LDS reads" : mode == 1 ? "reads one step ahead" : "reads one step ahead, one per MFMA gap", ms, flops / ms / 1e9,
                   ms * 1e-3 * 2.4e9 / (static_cast<double>(steps) * 8 * wps));
        }
    return 0;
}
