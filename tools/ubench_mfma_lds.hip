// ubench_mfma_lds.hip -- developer micro-benchmark (not part of the product; round 6): the rate of a v_mfma_f32_16x16x4_f32 stream whose B (and A) operands
// come from ds_read_b128 the way the fp32 kernels feed them (one 16-byte read per four MFMAs), against the same stream on constant registers, with 1 / 2 / 3
// waves per SIMD and 2 or 4 accumulator chains.  Asked by the irb_image ablation (tools/r6_iabl.sh): its MFMA skeleton runs at ~48 cycles per MFMA with one
// AND with two waves per SIMD, where tools/ubench_issue.hip (constant operands) gives 48 / 32.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma_lds.hip -o build/ubench_mfma_lds && build/ubench_mfma_lds
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: constant operands   1: B from LDS, read one group (4 MFMAs) ahead   2: B from LDS, read right before use   3: A and B from LDS, one group ahead
// 4: B from LDS one group ahead, accumulators forced into AGPRs is left to the compiler (same as 1) but 8 reads in flight
template <int MODE, int CH>
__global__ __launch_bounds__(256) void kern(float* out, long long* cyc, int iters) {
    extern __shared__ float4 sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = make_float4(i * 1e-3f, 1.f + i * 1e-4f, 0.5f, 0.25f);
    __syncthreads();
    f32x4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const float4* const bp = sm + wave * 1024 + lane;     // conflict-free: lane-contiguous 16-byte slots
    const float4* const ap = sm + wave * 1024 + 512 + lane;
    float4 bq = bp[0], aq = ap[0];
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) { // 8 groups of 4 MFMAs
            float4 bn = bq, an = aq;
            if (MODE == 1 || MODE == 3) bn = bp[((g + 1) & 7) * 64];
            if (MODE == 3) an = ap[((g + 1) & 7) * 64];
            if (MODE == 2) bq = bp[(g & 7) * 64];
            const float b0 = MODE == 0 ? b : bq.x, b1 = MODE == 0 ? b : bq.y, b2 = MODE == 0 ? b : bq.z, b3 = MODE == 0 ? b : bq.w;
            const float a0 = MODE == 3 ? aq.x : a, a1 = MODE == 3 ? aq.y : a, a2 = MODE == 3 ? aq.z : a, a3 = MODE == 3 ? aq.w : a;
            acc[(4 * g + 0) % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[(4 * g + 0) % CH], 0, 0, 0);
            acc[(4 * g + 1) % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[(4 * g + 1) % CH], 0, 0, 0);
            acc[(4 * g + 2) % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc[(4 * g + 2) % CH], 0, 0, 0);
            acc[(4 * g + 3) % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc[(4 * g + 3) % CH], 0, 0, 0);
            if (MODE == 1 || MODE == 3) bq = bn;
            if (MODE == 3) aq = an;
        }
    }
    const long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int CH>
void run(float* out, long long* cyc, int blocksPerCU) {
    const int iters = 4000, nb = 256 * blocksPerCU;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<MODE, CH>), dim3(nb), dim3(256), 65536, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<MODE, CH>), dim3(nb), dim3(256), 65536, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    static long long h[4096];
    hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < nb; ++i) s += h[i];
    const double mfmaPerSimd = static_cast<double>(iters) * 32 * blocksPerCU; // a block's wave w runs on one SIMD: blocksPerCU waves per SIMD
    const char* names[] = {"constant operands       ", "B from LDS, a group ahead", "B from LDS, before use  ", "A+B from LDS, group ahead"};
    printf("%s chains=%d waves/SIMD=%d: clock64 %.1f ticks per MFMA per wave | wall %.3f ms = %.1f ns per MFMA per SIMD (32 cycles at 2.4 GHz = 13.3 ns) -> %.1f TFLOP/s\n", names[MODE], CH,
           blocksPerCU, s / nb / (iters * 32.0), ms, ms * 1e6 / mfmaPerSimd, 2048.0 * mfmaPerSimd * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&cyc, 4096 * 8);
    for (int w = 1; w <= 2; ++w) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern<0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        run<0, 2>(out, cyc, w);
        run<0, 4>(out, cyc, w);
        run<0, 8>(out, cyc, w);
        run<1, 2>(out, cyc, w);
        run<1, 4>(out, cyc, w);
        run<1, 8>(out, cyc, w);
        run<2, 2>(out, cyc, w);
        run<2, 4>(out, cyc, w);
        run<3, 2>(out, cyc, w);
        run<3, 4>(out, cyc, w);
        run<3, 8>(out, cyc, w);
    }
    return 0;
}
