// ubench_hbm.hip -- developer microbenchmark (not part of the product): sustained read-only, write-only and copy bandwidth of one MI355X for
// 16-byte accesses, at a footprint that fits the 256 MB MALL and at ones that do not.  The write-only number is the floor of every layer
// whose output dominates its traffic (pointwise "expand" convolutions, the ESPCN c2 tensor).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_hbm.hip -o build/ubench_hbm
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

__global__ __launch_bounds__(256) void fill_kernel(float4* __restrict__ y, size_t n) {
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) y[i] = v;
}
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ x, size_t n, float* out) {
    float4 a = make_float4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
        const float4 v = x[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) *out = a.x;
}
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ x, float4* __restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) y[i] = x[i];
}
// one block per contiguous span (the access pattern of a tiled kernel's epilogue) instead of a grid-stride sweep
__global__ __launch_bounds__(256) void fill_span_kernel(float4* __restrict__ y, size_t n, size_t span) {
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    const size_t b = blockIdx.x * span;
    for (size_t i = threadIdx.x; i < span && b + i < n; i += 256) y[b + i] = v;
}

template <typename F>
static float timeit(F&& launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const size_t sizesMB[4] = {64, 160, 512, 2048};
    float* out;
    CK(hipMalloc(&out, 4));
    for (size_t mb : sizesMB) {
        const size_t bytes = mb << 20, n = bytes / 16;
        float4 *x, *y;
        CK(hipMalloc(&x, bytes));
        CK(hipMalloc(&y, bytes));
        CK(hipMemset(x, 0, bytes));
        for (int grid : {2048, 8192}) {
            const float tf = timeit([&] { hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, 0, y, n); }, 20);
            const float tr = timeit([&] { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, x, n, out); }, 20);
            const float tc = timeit([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, x, y, n); }, 20);
            printf("%5zu MB grid %5d: write-only %6.2f TB/s (%7.1f us) | read-only %6.2f TB/s (%7.1f us) | copy %6.2f TB/s r+w (%7.1f us)\n", mb, grid, bytes / tf / 1e9,
                   tf * 1e3, bytes / tr / 1e9, tr * 1e3, 2.0 * bytes / tc / 1e9, tc * 1e3);
        }
        const size_t span = 1024; // 16 KB per block
        const float ts = timeit([&] { hipLaunchKernelGGL(fill_span_kernel, dim3((unsigned) ((n + span - 1) / span)), dim3(256), 0, 0, y, n, span); }, 20);
        printf("%5zu MB one 16 KB span per block: write-only %6.2f TB/s (%7.1f us)\n", mb, bytes / ts / 1e9, ts * 1e3);
        CK(hipFree(x));
        CK(hipFree(y));
    }
    return 0;
}
