// tune_stream.hip -- developer harness for espcn_stream.hip (phase timing via s_memtime). Not part of the product.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifdef PHASE_TIMING
__device__ long long g_st[4096 * 16];
__device__ long long g_acc[4096 * 16];
#define SNNHIP_STREAM_STAMP(k)                                                              \
    do {                                                                                    \
        if ((threadIdx.x & 63) == 0) {                                                             \
            long long _t = clock64();                                                       \
            if ((k) > 0) g_acc[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (k)] += _t - g_st[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (k) -1]; \
            g_st[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (k)] = _t;                                                \
        }                                                                                   \
    } while (0)
#endif
#include "../shadernn_amd/csrc/espcn_stream.hip"
namespace snnhip { void set_error(const char*, ...) {} }
using namespace snnhip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
    const int H = 1080, W = 1920;
    std::vector<float> hx((size_t) H * W), hw(64 * 64);
    for (auto& v : hx) v = rand() / (float) RAND_MAX;
    for (auto& v : hw) v = rand() / (float) RAND_MAX - 0.5f;
    float *x, *w, *y;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&y, (size_t) H * W * 4 * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    alignas(8) char cfg[256];
    for (int cus : {256, 384, 512, 128}) {
        espcn_stream_configure(cfg, 1, H, W, 5, 1, 0.f, 1, 0.f, 0, 0.f, cus);
        char buf[300]; espcn_stream_describe(cfg, buf, sizeof(buf));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int i = 0; i < 3; ++i) espcn_stream_launch(0, cfg, x, w, w, w, w, w, w, y);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 30; ++i) espcn_stream_launch(0, cfg, x, w, w, w, w, w, w, y);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%.1f us  %s\n", ms * 1000 / 30, buf);
    }
#ifdef PHASE_TIMING
    for (int cus : {256, 128}) {
        espcn_stream_configure(cfg, 1, H, W, 5, 1, 0.f, 1, 0.f, 0, 0.f, cus);
        std::vector<long long> z(4096 * 16, 0);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_acc), z.data(), z.size() * 8));
        espcn_stream_launch(0, cfg, x, w, w, w, w, w, w, y);
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(z.data(), HIP_SYMBOL(g_acc), z.size() * 8));
        double s1 = 0, s2 = 0, s3 = 0; int nb = cus * 8;
        for (int b = 0; b < nb; ++b) { s1 += z[b * 4 + 1]; s2 += z[b * 4 + 2]; s3 += z[b * 4 + 3]; }
        printf("cus=%d waves=%d avg cycles per wave: conv1 %.0f | conv2 %.0f | c2epi+conv3+store+rotate %.0f\n", cus, nb, s1 / nb, s2 / nb, s3 / nb);
    }
#endif
    return 0;
}
