// tune_espcn.hip -- developer harness (not part of the product): times template variants of the fused ESPCN kernels
// on a 1080p frame with hipEvents.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_espcn.hip -o /tmp/tune
#include <hip/hip_runtime.h>
#ifdef PHASE_TIMING
__device__ long long g_stamps[8192 * 4 * 8];
__device__ long long g_wall[8192 * 4 * 2];
__device__ unsigned g_hwid[8192 * 4 * 2];
#define SNNHIP_STAMP(k)                                                                                          \
    do {                                                                                                         \
        if ((threadIdx.x & 63) == 0) {                                                                           \
            const int _w = blockIdx.x * 4 + (threadIdx.x >> 6);                                                  \
            g_stamps[_w * 8 + (k)] = clock64();                                                                  \
            if ((k) == 0) {                                                                                      \
                g_wall[_w * 2] = wall_clock64();                                                                 \
                g_hwid[_w * 2] = __builtin_amdgcn_s_getreg(0xF804);                                              \
                g_hwid[_w * 2 + 1] = __builtin_amdgcn_s_getreg(0xF814);                                          \
            }                                                                                                    \
            if ((k) == 6) g_wall[_w * 2 + 1] = wall_clock64();                                                   \
        }                                                                                                        \
    } while (0)
#endif
#include "../shadernn_amd/csrc/espcn_fused.hip"

#include <algorithm>
#include <map>
#include <cstdlib>
#include <vector>

using namespace snnhip;

namespace snnhip {
void set_error(const char* fmt, ...) { (void) fmt; }
std::vector<float> make_epilogue_table(int, int, int, const float*, int, const float*, const float*, const float*, const float*) { return {}; }
} // namespace snnhip
int snnhip_plan::upload(const float*, size_t, float**) { return 0; }
int snnhip_plan::profBegin(int) { return 0; }
int snnhip_plan::profEnd(int) { return 0; }
int snnhip_plan::profAcquire(int, hipEvent_t*, hipEvent_t*) { return 0; }
namespace snnhip {
int make_conv2d_mfma_plan(snnhip_ctx*, const ConvGeom&, const float*, const std::vector<float>&, snnhip_plan**) { return 0; }
bool instancenorm_plan_desc(const snnhip_plan*, snnhip_instancenorm_desc*) { return false; }
int instancenorm_apply_tile_stats(snnhip_plan*, const float*, int, int, int, int, snnhip_tensor*) { return 0; }
}
static snnhip::FusedBParams mkB(int H, int W, int tw, int th) {
    snnhip::FusedBParams p{1, H, W, (W + tw - 1) / tw, (H + th - 1) / th, snnhip::make_act_cfg(0, 0.f)};
    p.magicX = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(p.tilesX) - 1) / static_cast<unsigned>(p.tilesX));
    p.magicY = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(p.tilesY) - 1) / static_cast<unsigned>(p.tilesY));
    return p;
}
extern "C" int snnhip_tensor_alloc(snnhip_ctx*, int, int, int, int, int, snnhip_tensor**) { return 0; }
extern "C" int snnhip_tensor_free(snnhip_tensor*) { return 0; }
namespace snnhip {
size_t espcn_stream_step_size() { return 0; }
void espcn_stream_configure(void*, int, int, int, int, int, float, int, float, int, float, int) {}
void espcn_stream_describe(const void*, char*, size_t) {}
int espcn_stream_launch(hipStream_t, const void*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*) { return 0; }
}

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            printf("%s -> %s\n", #x, hipGetErrorString(e));                   \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

float timeBW(const float* x, const float* w, const float* e, float* y, int H, int W, int reps) {
    FusedBParams p = mkB(H, W, 64, 16);
    dim3 grid(512);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((conv3x3_c16o4_wino_d2s_tanh_kernel<true>), grid, dim3(256), 0, 0, p, x, w, e, y);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((conv3x3_c16o4_wino_d2s_tanh_kernel<true>), grid, dim3(256), 0, 0, p, x, w, e, y);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / reps;
}

// two-stream experiment: kernel A of image i+1 next to kernel B of image i (no dependency here: upper bound of what cross-step
// pipelining could give)
template <int BTW, int BTH>
float timeOverlap(const float* x, const float* w, float* mid, float* mid2, float* y, int H, int W, int reps, bool twoStreams) {
    FusedAParams pa{1, H, W, (W + WinoTile::TW - 1) / WinoTile::TW, (H + 15) / 16, make_act_cfg(1, 0.f), make_act_cfg(1, 0.f)};
    FusedBParams pb = mkB(H, W, BTW, BTH);
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    hipEvent_t a, b, e2;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipEventCreate(&e2));
    auto once = [&](int i) {
        hipLaunchKernelGGL((conv_kxk_c1o16_wino3x3_c16o16_kernel<5, 16, 2, 2>), dim3(512), dim3(256), 0, s1, pa, x, w, w, w, w, (i & 1) ? mid2 : mid);
        hipLaunchKernelGGL((conv3x3_c16o4_d2s_tanh_kernel<BTW, BTH, true>), dim3(pb.tilesX * pb.tilesY), dim3(256), 0, twoStreams ? s2 : s1, pb,
                           (i & 1) ? mid : mid2, w, w, y);
    };
    for (int i = 0; i < 6; ++i) once(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, s1));
    for (int i = 0; i < reps; ++i) once(i);
    CK(hipEventRecord(e2, s2));
    CK(hipStreamWaitEvent(s1, e2, 0));
    CK(hipEventRecord(b, s1));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / reps;
}

float timeBD(const float* x, const float* w, const float* e, const float* zeros, float* y, int H, int W, int reps, int blocksPerCU) {
    FusedBParams p = mkB(H, W, 32, 8);
    dim3 grid(256 * blocksPerCU);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((conv3x3_c16o4_d2s_tanh_dma_kernel<true>), grid, dim3(256), 0, 0, p, x, w, e, zeros, y);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((conv3x3_c16o4_d2s_tanh_dma_kernel<true>), grid, dim3(256), 0, 0, p, x, w, e, zeros, y);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / reps;
}

template <int K1, int WTH, int WWPS>
float timeW(const float* x, const float* w1, const float* w2, const float* e1, const float* e2, float* y, int H, int W, int reps) {
    FusedAParams p{1, H, W, (W + WinoTile::TW - 1) / WinoTile::TW, (H + WTH - 1) / WTH, make_act_cfg(1, 0.f), make_act_cfg(1, 0.f)};
    dim3 grid(256 * WWPS);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((conv_kxk_c1o16_wino3x3_c16o16_kernel<K1, WTH, 2, WWPS>), grid, dim3(256), 0, 0, p, x, w1, w2, e1, e2, y);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((conv_kxk_c1o16_wino3x3_c16o16_kernel<K1, WTH, 2, WWPS>), grid, dim3(256), 0, 0, p, x, w1, w2, e1, e2, y);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / reps;
}

template <int K1, int TW, int TH, int U = 3, int WPS = 2>
float timeA(const float* x, const float* w1, const float* w2, const float* e1, const float* e2, float* y, int H, int W, int reps) {
    FusedAParams p{1, H, W, (W + TW - 1) / TW, (H + TH - 1) / TH, make_act_cfg(1, 0.f), make_act_cfg(1, 0.f)};
    dim3 grid(p.tilesX * p.tilesY);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv_kxk_c1o16_conv3x3_c16o16_kernel<K1, TW, TH, true, U, WPS>), grid, dim3(256), 0, 0, p, x, w1, w2, e1, e2, y);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((conv_kxk_c1o16_conv3x3_c16o16_kernel<K1, TW, TH, true, U, WPS>), grid, dim3(256), 0, 0, p, x, w1, w2, e1, e2, y);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / reps;
}

template <int TW, int TH>
float timeB(const float* x, const float* w, const float* e, float* y, int H, int W, int reps) {
    FusedBParams p = mkB(H, W, TW, TH);
    dim3 grid(p.tilesX * p.tilesY);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv3x3_c16o4_d2s_tanh_kernel<TW, TH, true>), grid, dim3(256), 0, 0, p, x, w, e, y);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((conv3x3_c16o4_d2s_tanh_kernel<TW, TH, true>), grid, dim3(256), 0, 0, p, x, w, e, y);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / reps;
}

int main() {
    const int H = 1080, W = 1920;
    std::vector<float> hx((size_t) H * W), hw(64 * 64);
    for (auto& v : hx) v = rand() / (float) RAND_MAX;
    for (auto& v : hw) v = rand() / (float) RAND_MAX - 0.5f;
    float *x, *w, *mid, *y;
    CK(hipMalloc(&x, hx.size() * 4));
    CK(hipMalloc(&w, hw.size() * 4));
    CK(hipMalloc(&mid, (size_t) H * W * 16 * 4));
    CK(hipMalloc(&y, (size_t) H * W * 4 * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    const int R = 50;
    {
        hipFuncAttributes fa;
        auto kfn = conv_kxk_c1o16_conv3x3_c16o16_kernel<5, 64, 8, true, 3, 3>;
        CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kfn)));
        int nb = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, 256, 0));
        hipDeviceProp_t pr;
        CK(hipGetDeviceProperties(&pr, 0));
        printf("A<5,64,8>: numRegs %d sharedSizeBytes %zu maxDynShared %d | occupancy API blocks/CU %d | LDS/CU %zu maxShared/block %zu regs/CU %d\n", fa.numRegs,
               fa.sharedSizeBytes, fa.maxDynamicSharedSizeBytes, nb, pr.maxSharedMemoryPerMultiProcessor, pr.sharedMemPerBlock, pr.regsPerMultiprocessor);
    }
#ifdef PHASE_TIMING
    {
#ifdef PHASE_BW
        FusedBParams p = mkB(H, W, 64, 16);
        int nb = p.tilesX * p.tilesY;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((conv_kxk_c1o16_wino3x3_c16o16_kernel<5, 16, 2, 2>), dim3(512), dim3(256), 0, 0,
                               FusedAParams{1, H, W, 60, 68, make_act_cfg(1, 0.f), make_act_cfg(1, 0.f)}, x, w, w, w, w, mid);
            hipLaunchKernelGGL((conv3x3_c16o4_wino_d2s_tanh_kernel<true>), dim3(512), dim3(256), 0, 0, p, mid, w, w, y);
            nb = 512;
            CK(hipDeviceSynchronize());
        }
#elif defined(PHASE_WINO)
        constexpr int WTH = PHASE_WINO_TH, WWPS = PHASE_WINO_WPS;
        FusedAParams p{1, H, W, (W + WinoTile::TW - 1) / WinoTile::TW, (H + WTH - 1) / WTH, make_act_cfg(1, 0.f), make_act_cfg(1, 0.f)};
        int nb = 256 * WWPS;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((conv_kxk_c1o16_wino3x3_c16o16_kernel<5, WTH, 2, WWPS>), dim3(nb), dim3(256), 0, 0, p, x, w, w, w, w, mid);
            CK(hipDeviceSynchronize());
        }
#else
        FusedAParams p{1, H, W, (W + 63) / 64, (H + 7) / 8, make_act_cfg(1, 0.f), make_act_cfg(1, 0.f)};
        int nb = p.tilesX * p.tilesY;
        hipLaunchKernelGGL((conv_kxk_c1o16_conv3x3_c16o16_kernel<5, 64, 8, true, 3, 3>), dim3(nb), dim3(256), 0, 0, p, x, w, w, w, w, mid);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL((conv_kxk_c1o16_conv3x3_c16o16_kernel<5, 64, 8, true, 3, 3>), dim3(nb), dim3(256), 0, 0, p, x, w, w, w, w, mid);
        CK(hipDeviceSynchronize());
#endif
        std::vector<long long> st((size_t) 8192 * 4 * 8);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
        long long tmin = st[0];
        for (int b = 0; b < nb; ++b) for (int w4 = 0; w4 < 4; ++w4) tmin = std::min(tmin, st[(b * 4 + w4) * 8]);
        double sum[7] = {0};
        for (int b = 0; b < nb; ++b) for (int w4 = 0; w4 < 4; ++w4) for (int k = 1; k < 7; ++k) sum[k] += st[(b * 4 + w4) * 8 + k] - st[(b * 4 + w4) * 8 + k - 1];
        printf("avg ticks per wave: load %.0f | bar %.0f | conv1 %.0f | bar %.0f | conv2 %.0f | epi %.0f\n", sum[1] / (nb * 4), sum[2] / (nb * 4), sum[3] / (nb * 4), sum[4] / (nb * 4), sum[5] / (nb * 4), sum[6] / (nb * 4));
        std::vector<long long> wl((size_t) 8192 * 4 * 2);
        std::vector<unsigned> hw((size_t) 8192 * 4 * 2);
        CK(hipMemcpyFromSymbol(wl.data(), HIP_SYMBOL(g_wall), wl.size() * 8));
        CK(hipMemcpyFromSymbol(hw.data(), HIP_SYMBOL(g_hwid), hw.size() * 4));
        long long w0 = wl[0], w1 = 0;
        for (int b = 0; b < nb; ++b) { w0 = std::min(w0, wl[(b * 4) * 2]); w1 = std::max(w1, wl[(b * 4) * 2 + 1]); }
        printf("kernel wall span %.2f us (100 MHz ticks)\n", (w1 - w0) / 100.0);
        // cycles per us from one block
        double cyc = st[6] - st[0], us = (wl[1] - wl[0]) / 100.0;
        printf("block0: %.0f cycles in %.2f us -> %.3f GHz\n", cyc, us, cyc / us / 1e3);
        // residency: sum of block wall durations / (span * CUs)
        double tot = 0;
        for (int b = 0; b < nb; ++b) tot += (wl[(b * 4) * 2 + 1] - wl[(b * 4) * 2]) / 100.0;
        printf("sum of block lifetimes %.1f us -> avg %.2f blocks resident per CU (256 CUs)\n", tot, tot / ((w1 - w0) / 100.0) / 256);
        // per-CU histogram: key = xcc, se, sh, cu
        std::map<unsigned, int> perCU;
        for (int b = 0; b < nb; ++b) {
            unsigned h = hw[(b * 4) * 2], x = hw[(b * 4) * 2 + 1] & 0xF;
            unsigned key = (x << 16) | (((h >> 13) & 7) << 8) | (((h >> 12) & 1) << 4) | ((h >> 8) & 15);
            perCU[key]++;
        }
        int mn = 1 << 30, mx = 0;
        for (auto& kv : perCU) { mn = std::min(mn, kv.second); mx = std::max(mx, kv.second); }
        printf("distinct CUs seen %zu, blocks per CU min %d max %d\n", perCU.size(), mn, mx);
        // start-time histogram of first 1000 blocks
        for (int b : {0, 255, 256, 511, 512, 767, 768, 1023, 1024, 1500}) printf("  block %d start +%.2f us end +%.2f us\n", b, (wl[(b * 4) * 2] - w0) / 100.0, (wl[(b * 4) * 2 + 1] - w0) / 100.0);
        for (int b : {0, 1, 2, 767, 768, 769, 2000, 4049}) {
            printf("block %d wave0: start %lld", b, st[(b * 4) * 8] - tmin);
            for (int k = 1; k < 7; ++k) printf(" +%lld", st[(b * 4) * 8 + k] - st[(b * 4) * 8 + k - 1]);
            printf("\n");
        }
    }
#endif
#ifdef TUNE_VARIANTS
    TUNE_VARIANTS
#else
    {
        float* mid2;
        CK(hipMalloc(&mid2, (size_t) H * W * 16 * 4));
        CK(hipMemset(mid2, 0, (size_t) H * W * 16 * 4));
        printf("A+B one stream  %.1f us/step\n", timeOverlap<32, 8>(x, w, mid, mid2, y, H, W, R, false));
        printf("A|B two streams %.1f us/step\n", timeOverlap<32, 8>(x, w, mid, mid2, y, H, W, R, true));
    }
    printf("W<5> wino  %.1f us\n", timeW<5, 16, 2>(x, w, w, w, w, mid, H, W, R));
    printf("W<5,8,3> wino  %.1f us\n", timeW<5, 8, 3>(x, w, w, w, w, mid, H, W, R));
    printf("A<5,64,8,U3,W3>  %.1f us\n", timeA<5, 64, 8, 3, 3>(x, w, w, w, w, mid, H, W, R));
    printf("BW wino    %.1f us\n", timeBW(mid, w, w, y, H, W, R));
    {
        float* zeros;
        CK(hipMalloc(&zeros, 256));
        CK(hipMemset(zeros, 0, 256));
        for (int bpc : {2, 3}) printf("BD dma persistent x%d %.1f us\n", bpc, timeBD(mid, w, w, zeros, y, H, W, R, bpc));
    }
    printf("B<32,8>    %.1f us\n", timeB<32, 8>(mid, w, w, y, H, W, R));
    printf("B<64,4>    %.1f us\n", timeB<64, 4>(mid, w, w, y, H, W, R));
    printf("B<16,16>   %.1f us\n", timeB<16, 16>(mid, w, w, y, H, W, R));
#endif
    return 0;
}
