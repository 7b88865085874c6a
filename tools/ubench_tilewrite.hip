// ubench_tilewrite.hip -- developer microbenchmark (not part of the product): write-only bandwidth of one MI355X when an NHWC tensor is written TILE by tile the way the
// convolution kernels' epilogues do it (conv2d_stem_f16: 512 persistent blocks, a 256-thread block owns a TH x TW pixel tile of 64-byte pixels, a wave store instruction
// covers 16 adjacent pixels = 1 KB, the tile's rows are a whole image row apart), against the same bytes written as one contiguous sweep.  Round 6: the stem's ablation
// builds and phase trace say its pace is set by the memory system at ~2.3 TB/s of writes; this asks what the pattern alone can reach.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_tilewrite.hip -o build/ubench_tilewrite
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

// pixel = 64 bytes = 4 float4; a thread stores one float4: lane L = piece L % 4 of pixel L / 4 (+ 16 per pass): 1 KB contiguous per wave instruction
__global__ __launch_bounds__(256) void tile_write(float4* __restrict__ y, int N, int OH, int OW, int TH, int TW, int contiguousOrder) {
    const int tilesX = OW / TW, tilesY = OH / TH, tpi = tilesX * tilesY, total = tpi * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    const int chunk = (total + gridDim.x - 1) / gridDim.x;
    for (int i = 0; i < chunk; ++i) {
        const int mt = contiguousOrder ? blockIdx.x * chunk + i : blockIdx.x + i * gridDim.x; // a block's tiles: a contiguous run, or strided by the grid
        if (mt >= total) break;
        const int n = mt / tpi, ty = (mt / tilesX) % tilesY, tx = mt % tilesX;
        for (int r = wave; r < TH; r += 4) { // the wave's rows of the tile
            const size_t rowBase = ((static_cast<size_t>(n) * OH + ty * TH + r) * OW + tx * TW) * 4; // float4 index
            for (int px = 0; px < TW; px += 16) y[rowBase + static_cast<size_t>(px) * 4 + lane] = v;
        }
    }
}
__global__ __launch_bounds__(256) void fill_kernel(float4* __restrict__ y, size_t n) {
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) y[i] = v;
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
    const int N = 16, OH = 720, OW = 1280; // Candy's stem output: 16 x 720 x 1280 x 32 halfs = 944 MB
    const size_t n4 = static_cast<size_t>(N) * OH * OW * 4, bytes = n4 * 16;
    float4* y;
    CK(hipMalloc(&y, bytes));
    const float tf = timeit([&] { hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, y, n4); }, 10);
    printf("%zu MB contiguous grid-stride fill: %6.2f TB/s (%7.1f us)\n", bytes >> 20, bytes / tf / 1e9, tf * 1e3);
    const int shapes[6][2] = {{16, 32}, {8, 64}, {4, 128}, {16, 64}, {8, 128}, {4, 1280}};
    for (auto& sh : shapes)
        for (int order = 0; order < 2; ++order)
            for (int grid : {512, 1024}) {
                const float t = timeit([&] { hipLaunchKernelGGL(tile_write, dim3(grid), dim3(256), 0, 0, y, N, OH, OW, sh[0], sh[1], order); }, 10);
                printf("tile %2d rows x %4d px (%5d B per row), %s tile order, grid %4d: %6.2f TB/s (%7.1f us)\n", sh[0], sh[1], sh[1] * 64, order ? "contiguous" : "grid-strided", grid,
                       bytes / t / 1e9, t * 1e3);
            }
    CK(hipFree(y));
    return 0;
}
