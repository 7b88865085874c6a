// eltwise_pool.hip -- the element-wise, pooling and shape operators that sit between the convolutions of a ResNet-18 /
// MobileNetV2 / Candy graph.  All are HBM-bound: NHWC fp32, 16-byte channel-contiguous
// accesses whenever C % 4 == 0 (scalar path otherwise), grid-stride loops sized to a few waves per SIMD.
//
// Replaces (core/data/assets/shaders): shadertemplate_vk_add.comp:41-88, vk_activation.comp:41-86, vk_batchnorm.comp:54-104,
// vk_maxpool2d.comp:42-74, vk_avgpool2d.comp:42-69, vk_pad.comp:42-71, vk_upsampling2d_nearest.comp:43-66,
// vk_upsampling2d_bilinear.comp:43-76, vk_instancenorm.comp:53-160 and their createCS hosts in core/src/ic2.
#include "epilogue.h"
#include "plan_util.h"
#include "snnhip_internal.h"

namespace snnhip {
namespace {

__device__ __forceinline__ float act1(int act, float leaky, float v) { return epi_act(act, leaky, v, 0.0f); }
__device__ __forceinline__ float4 act4(int act, float leaky, float4 v) {
    return make_float4(act1(act, leaky, v.x), act1(act, leaky, v.y), act1(act, leaky, v.z), act1(act, leaky, v.w));
}

// ------------------------------------------------------------------------------------------------ add / activation / batch-norm
// mode 0: y = act(a + b)   mode 1: y = act(a)   mode 2: y = act(scale[c] * (a - mean[c]) + beta[c]),  tab[c] = {scale, mean, beta, 0}
template <int MODE, int CV, typename T>
__global__ __launch_bounds__(256) void eltwise_kernel(size_t count, int C, int act, float leaky, const T* __restrict__ a, const T* __restrict__ b,
                                                     const float4* __restrict__ tab, T* __restrict__ y) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    const size_t ng = count / CV;
    const int cg = C / CV;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < ng; i += stride) {
        float v[CV];
        ldv<T, CV>(a + i * CV, v);
        if (MODE == 0) {
            float w[CV];
            ldv<T, CV>(b + i * CV, w);
#pragma unroll
            for (int k = 0; k < CV; ++k) v[k] += w[k];
        }
        if (MODE == 2) {
            const int c = static_cast<int>(i % cg) * CV;
#pragma unroll
            for (int k = 0; k < CV; ++k) {
                const float4 t = tab[c + k];
                v[k] = t.x * (v[k] - t.y) + t.z;
            }
        }
#pragma unroll
        for (int k = 0; k < CV; ++k) v[k] = act1(act, leaky, v[k]);
        stv<T, CV>(y + i * CV, v);
    }
}

// Add with inputs of different spatial size (same N, C).  The reference sizes the output as the MAX over its inputs
// (genericlayer.cpp:64-90) but dispatches the sum only over the FIRST input's extent, and the second input's out-of-range fetches
// return 0 (addlayerVulkan.cpp:44-46,89-91; vk_add.comp:47-49).  Candy's residual blocks rely on this: a Pad + "valid" conv branch is
// 4 pixels larger than its skip under the reference's size rule.  Outside the first input's extent the reference leaves
// the texture untouched (undefined); zeros are written here.
template <int CV, typename T>
__global__ __launch_bounds__(256) void add_ragged_kernel(int N, int H, int W, int C, int H0, int W0, int H1, int W1, int act, float leaky,
                                                        const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y) {
    const int cg = C / CV;
    const size_t total = static_cast<size_t>(N) * H * W * cg;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const int c = static_cast<int>(i % cg) * CV;
        size_t r = i / cg;
        const int ox = static_cast<int>(r % W);
        r /= W;
        const int oy = static_cast<int>(r % H);
        const int n = static_cast<int>(r / H);
        const bool in0 = oy < H0 && ox < W0, in1 = oy < H1 && ox < W1;
        float v[CV];
#pragma unroll
        for (int k = 0; k < CV; ++k) v[k] = 0.0f;
        if (in0) {
            ldv<T, CV>(a + ((static_cast<size_t>(n) * H0 + oy) * W0 + ox) * C + c, v);
            if (in1) {
                float w[CV];
                ldv<T, CV>(b + ((static_cast<size_t>(n) * H1 + oy) * W1 + ox) * C + c, w);
#pragma unroll
                for (int k = 0; k < CV; ++k) v[k] += w[k];
            }
#pragma unroll
            for (int k = 0; k < CV; ++k) v[k] = act1(act, leaky, v[k]);
        }
        stv<T, CV>(y + ((static_cast<size_t>(n) * H + oy) * W + ox) * C + c, v);
    }
}

// ------------------------------------------------------------------------------------------------ pooling
// one thread = one output pixel x CV channels; window clipped to the image exactly as the shader does
template <int TYPE, int CV, typename T>
__global__ __launch_bounds__(256) void pool2d_kernel(snnhip_pool2d_desc d, const T* __restrict__ x, T* __restrict__ y) {
    const int cg = d.C / CV;
    const size_t total = static_cast<size_t>(d.N) * d.OH * d.OW * cg;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const int c = static_cast<int>(i % cg) * CV;
        size_t r = i / cg;
        const int ox = static_cast<int>(r % d.OW);
        r /= d.OW;
        const int oy = static_cast<int>(r % d.OH);
        const int n = static_cast<int>(r / d.OH);
        const int sx = ox * d.sw - d.padL, sy = oy * d.sh - d.padT;
        const int fx0 = max(0, -sx), fy0 = max(0, -sy);
        const int fx1 = min(d.kw, d.W - sx), fy1 = min(d.kh, d.H - sy);
        float acc[CV];
#pragma unroll
        for (int k = 0; k < CV; ++k) acc[k] = TYPE == SNNHIP_POOL_MAX ? -100000.0f : 0.0f;
        float num = 0.0f;
        for (int fy = fy0; fy < fy1; ++fy)
            for (int fx = fx0; fx < fx1; ++fx) {
                float v[CV];
                ldv<T, CV>(x + ((static_cast<size_t>(n) * d.H + sy + fy) * d.W + sx + fx) * d.C + c, v);
#pragma unroll
                for (int k = 0; k < CV; ++k) acc[k] = TYPE == SNNHIP_POOL_MAX ? fmaxf(acc[k], v[k]) : acc[k] + v[k];
                num += 1.0f;
            }
        if (TYPE == SNNHIP_POOL_AVG) {
#pragma unroll
            for (int k = 0; k < CV; ++k) acc[k] = acc[k] / num; // an empty window divides 0 by 0 like the shader (vk_avgpool2d.comp:66)
        }
        stv<T, CV>(y + ((static_cast<size_t>(n) * d.OH + oy) * d.OW + ox) * d.C + c, acc);
    }
}

// Whole-image average (AdaptiveAvgPool2d / a pooling window that covers the map: the classifier heads of ResNet-18 and MobileNetV2).  One thread
// per output of pool2d_kernel walks the H*W pixels with one dependent load each (17 us for 32x7x7x512: pure latency); here 16 lanes share an
// output quad, each sums every 16th pixel, and a 4-step xor-shuffle adds them up.  Same result up to the order of the fp32 additions.
template <typename T>
__global__ __launch_bounds__(256) void global_avgpool_kernel(int N, int HW, int C, const T* __restrict__ x, T* __restrict__ y) {
    const int cg = C >> 2;
    const int part = threadIdx.x & 15;
    const size_t total = static_cast<size_t>(N) * cg; // outputs (channel quads)
    const size_t o = static_cast<size_t>(blockIdx.x) * 16 + (threadIdx.x >> 4);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (o < total) {
        const int n = static_cast<int>(o / cg), c = static_cast<int>(o % cg) * 4;
        const T* px = x + static_cast<size_t>(n) * HW * C + c;
        for (int i = part; i < HW; i += 16) {
            float v[4];
            ldv<T, 4>(px + static_cast<size_t>(i) * C, v);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += v[k];
        }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += __shfl_xor(acc[k], m, 64);
    if (o < total && part == 0) {
        const float num = static_cast<float>(HW);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = acc[k] / num;
        stv<T, 4>(y + o * 4, acc);
    }
}

// ------------------------------------------------------------------------------------------------ pad
template <int CV, typename T>
__global__ __launch_bounds__(256) void pad_kernel(snnhip_pad_desc d, int OH, int OW, const T* __restrict__ x, T* __restrict__ y) {
    const int cg = d.C / CV;
    const size_t total = static_cast<size_t>(d.N) * OH * OW * cg;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const int c = static_cast<int>(i % cg) * CV;
        size_t r = i / cg;
        const int ox = static_cast<int>(r % OW);
        r /= OW;
        const int oy = static_cast<int>(r % OH);
        const int n = static_cast<int>(r / OH);
        int sx = ox - d.padT, sy = oy - d.padL; // sic: x shifted by the TOP pad, y by the LEFT pad (padlayerVulkan.cpp:81-82)
        bool zero = false;
        if (d.mode == 0) {
            zero = !(sx >= 0 && sx < d.W && sy >= 0 && sy < d.H); // the shader fetches texel (W, H): out of range -> 0
        } else if (d.mode == 1) {
            sx = min(max(sx, 0), d.W - 1);
            sy = min(max(sy, 0), d.H - 1);
        } else {
            sx = sx < 0 ? -sx : sx;
            sx = sx >= d.W ? 2 * d.W - 2 - sx : sx;
            sy = sy < 0 ? -sy : sy;
            sy = sy >= d.H ? 2 * d.H - 2 - sy : sy;
            zero = !(sx >= 0 && sx < d.W && sy >= 0 && sy < d.H); // pads wider than the image leave the texture range
        }
        float v[CV];
#pragma unroll
        for (int k = 0; k < CV; ++k) v[k] = 0.0f;
        if (!zero) ldv<T, CV>(x + ((static_cast<size_t>(n) * d.H + sy) * d.W + sx) * d.C + c, v);
        stv<T, CV>(y + ((static_cast<size_t>(n) * OH + oy) * OW + ox) * d.C + c, v);
    }
}

// ------------------------------------------------------------------------------------------------ upsampling
template <int MODE, int CV, typename T>
__global__ __launch_bounds__(256) void upsample_kernel(snnhip_upsample_desc d, int OH, int OW, float inv, const T* __restrict__ x, T* __restrict__ y) {
    const int cg = d.C / CV;
    const size_t total = static_cast<size_t>(d.N) * OH * OW * cg;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const int c = static_cast<int>(i % cg) * CV;
        size_t r = i / cg;
        const int ox = static_cast<int>(r % OW);
        r /= OW;
        const int oy = static_cast<int>(r % OH);
        const int n = static_cast<int>(r / OH);
        const T* xn = x + static_cast<size_t>(n) * d.H * d.W * d.C + c;
        float o[CV];
        auto fetch = [&](int px, int py, float (&v)[CV]) { // texelFetch outside the texture returns 0
            const bool ok = px >= 0 && px < d.W && py >= 0 && py < d.H;
#pragma unroll
            for (int k = 0; k < CV; ++k) v[k] = 0.0f;
            if (ok) ldv<T, CV>(xn + (static_cast<size_t>(py) * d.W + px) * d.C, v);
        };
        if (MODE == SNNHIP_UPSAMPLE_NEAREST) {
            const int x1 = min(max(static_cast<int>(floorf(static_cast<float>(ox) * inv)), 0), d.W - 1);
            const int y1 = min(max(static_cast<int>(floorf(static_cast<float>(oy) * inv)), 0), d.H - 1);
            fetch(x1, y1, o);
        } else {
            const float off = 0.5f - 0.5f * inv;
            float srcX = static_cast<float>(ox) * inv - off;
            srcX = fminf(fmaxf(srcX, 0.0f), static_cast<float>(d.W - 1));
            const int x11 = static_cast<int>(floorf(srcX)), x12 = x11 + 1;
            float srcY = static_cast<float>(oy) * inv - off;
            srcY = fminf(fmaxf(srcY, 0.0f), static_cast<float>(d.H - 1));
            const int y11 = static_cast<int>(floorf(srcY)), y12 = y11 + 1;
            float r1[CV], r2[CV], r3[CV], r4[CV];
            fetch(x11, y11, r1);
            fetch(x12, y11, r2);
            fetch(x12, y12, r3);
            fetch(x11, y12, r4);
            const float w1 = (static_cast<float>(x12) - srcX) * (static_cast<float>(y12) - srcY);
            const float w2 = (srcX - static_cast<float>(x11)) * (static_cast<float>(y12) - srcY);
            const float w3 = (srcX - static_cast<float>(x11)) * (srcY - static_cast<float>(y11));
            const float w4 = (static_cast<float>(x12) - srcX) * (srcY - static_cast<float>(y11));
#pragma unroll
            for (int k = 0; k < CV; ++k) o[k] = r1[k] * w1 + r2[k] * w2 + r3[k] * w3 + r4[k] * w4;
        }
        stv<T, CV>(y + ((static_cast<size_t>(n) * OH + oy) * OW + ox) * d.C + c, o);
    }
}

// ------------------------------------------------------------------------------------------------ instance norm
// Statistics per (image, channel) over H*W.  The shader makes two passes (mean, then sum (x-mean)^2); here ONE statistics sweep
// accumulates S1 = sum (x - p) and S2 = sum (x - p)^2 around a per-channel pivot p = x[n, 0, 0, c] (so that S2/HW - (S1/HW)^2 does not
// cancel: the pivot is within a few standard deviations of the mean), then one sweep normalises: 2 reads + 1 write of the tensor
// instead of 3 + 1.  Work split: block = (image n, row slab s); its 256 threads are CL channel lanes x 256/CL pixel lanes with
// CL*CV >= min(C, 128) so that one wave instruction reads whole pixels (contiguous C*4 bytes) instead of half-lines.
// Partials go to part[n][s][2][C]; a tiny fold kernel turns them into mean[n][C] and mul[n][C] in a fixed order (deterministic).
// STAGE 0: partial S1, S2      STAGE 2: y = act((x - mean) * mul + beta)
struct InResidual { // graph rule H: the Add behind an InstanceNorm, applied in the norm's normalise sweep
    const void* p = nullptr;
    int H = 0, W = 0;
    int zeroOutside = 0; // the residual is the Add's first input: the reference dispatches over ITS extent only (zeros elsewhere, see add_ragged_kernel)
    int act = 0;
    float leaky = 0.0f;
};

template <int STAGE, int CV, typename T, int FAST = 0 /* STAGE 2, the Add's activation none: 1 = the norm's activation is ReLU (max(x, 0)), 2 = none (no max at all: a NaN stays a NaN) -- no run-time switch per value */>
__global__ __launch_bounds__(256) void instancenorm_kernel(snnhip_instancenorm_desc d, int S, int pixelsPerSlab, int CLs, const T* __restrict__ x,
                                                          const float* __restrict__ statMean, const float* __restrict__ statMul,
                                                          const float* __restrict__ beta, float* __restrict__ partOut, T* __restrict__ y,
                                                          InResidual ra = InResidual()) {
    // ra.p != nullptr (STAGE 2, graph rule H): the Add layer behind the norm is applied in the same sweep, y = addAct(T(act(norm(x))) + res) with the
    // rounding point of the separate launches (the norm's result is rounded to the tensor type before the addition).  The residual may be smaller than
    // the norm (top-left aligned, add_ragged_kernel's rule): outside it the sum is the norm alone, or 0 when the residual is the Add's FIRST input.
    const T* __restrict__ res = static_cast<const T*>(ra.p);
    const bool ragged = ra.p && (ra.H != d.H || ra.W != d.W);
    __shared__ float red[2 * 256 * CV];
    const int n = blockIdx.x / S, s = blockIdx.x % S;
    const int tid = threadIdx.x;
    const int CL = 1 << CLs, PL = 256 >> CLs;       // channel lanes, pixel lanes
    const int cl = tid & (CL - 1), pl = tid >> CLs;
    const size_t HW = static_cast<size_t>(d.H) * d.W;
    const size_t p0 = static_cast<size_t>(s) * pixelsPerSlab, p1 = min(HW, p0 + pixelsPerSlab); // slab = a pixel range of the image
    const T* xn = x + static_cast<size_t>(n) * HW * d.C;
    T* yn = y + static_cast<size_t>(n) * HW * d.C;
    // independent loads in flight per thread: the sweeps are latency-bound otherwise (measured 0.9 TB/s with one).  More is not better: 8 in flight
    // ran the fp16 normalise + Add sweep at 0.55x (400 vs 220 us for 3 x 285 MB), and so did 16-byte accesses (8 halfs per thread: 0.73x)
    constexpr int U = 4;
    for (int c0 = 0; c0 < d.C; c0 += CL * CV) {
        const int c = c0 + cl * CV;
        const bool cok = c < d.C;
        float piv[CV], mul[CV], bt[CV];
#pragma unroll
        for (int k = 0; k < CV; ++k) {
            piv[k] = cok ? (STAGE == 0 ? static_cast<float>(xn[c + k]) : statMean[static_cast<size_t>(n) * d.C + c + k]) : 0.0f; // stage 0: pivot, stage 2: shift
            mul[k] = (STAGE == 2 && cok) ? statMul[static_cast<size_t>(n) * d.C + c + k] : 0.0f;
            bt[k] = (STAGE == 2 && cok) ? beta[c + k] : 0.0f;
        }
        float s1[CV], s2[CV];
#pragma unroll
        for (int k = 0; k < CV; ++k) s1[k] = s2[k] = 0.0f;
        // residual of pixel p (rule H): where it sits in the (possibly smaller) residual tensor, or -1 outside it.  32-bit arithmetic: H * W < 2^31.
        auto res_pixel = [&](size_t p) -> long {
            if (!ragged) return static_cast<long>(p);
            const unsigned pu = static_cast<unsigned>(p), oy = pu / static_cast<unsigned>(d.W), ox = pu - oy * static_cast<unsigned>(d.W);
            return (static_cast<int>(oy) < ra.H && static_cast<int>(ox) < ra.W) ? static_cast<long>(oy) * ra.W + ox : -1;
        };
        auto consume = [&](const float (&v)[CV], size_t p, const float (&rv)[CV], long rp) {
            if (STAGE == 0) {
#pragma unroll
                for (int k = 0; k < CV; ++k) {
                    const float dv = v[k] - piv[k];
                    s1[k] += dv;
                    s2[k] += dv * dv;
                }
            } else {
                float o[CV];
                if (FAST) { // (epi_act's switch per value -- 16 values in flight per thread, each behind its own branches -- is what the generic form below costs)
#pragma unroll
                    for (int k = 0; k < CV; ++k) {
                        const float f = fmaf(v[k], mul[k], piv[k]);
                        o[k] = FAST == 1 ? fmaxf(f, 0.0f) : f;
                    }
                    if (res) {
                        if (rp >= 0) {
#pragma unroll
                            for (int k = 0; k < CV; ++k) o[k] = static_cast<float>(static_cast<T>(o[k])) + rv[k];
                        } else if (ra.zeroOutside) {
#pragma unroll
                            for (int k = 0; k < CV; ++k) o[k] = 0.0f;
                        }
                    }
                } else {
#pragma unroll
                for (int k = 0; k < CV; ++k) o[k] = act1(d.act, d.leaky, fmaf(v[k], mul[k], piv[k])); // x * mul + shift (shift = beta - mean * mul, from the fold)
                if (res) {
                    if (rp >= 0) {
#pragma unroll
                        for (int k = 0; k < CV; ++k) o[k] = act1(ra.act, ra.leaky, static_cast<float>(static_cast<T>(o[k])) + rv[k]);
                    } else {
#pragma unroll
                        for (int k = 0; k < CV; ++k) o[k] = ra.zeroOutside ? 0.0f : act1(ra.act, ra.leaky, static_cast<float>(static_cast<T>(o[k])));
                    }
                }
                }
                stv<T, CV>(yn + p * d.C + c, o);
            }
        };
        const T* const resn = (STAGE == 2 && res) ? res + static_cast<size_t>(n) * ra.H * ra.W * d.C + c : nullptr;
        if (cok) {
            size_t p = p0 + pl;
            for (; p + static_cast<size_t>(U - 1) * PL < p1; p += static_cast<size_t>(U) * PL) {
                float v[U][CV], rv[U][CV];
                long rp[U];
#pragma unroll
                for (int u = 0; u < U; ++u) ldv<T, CV>(xn + (p + static_cast<size_t>(u) * PL) * d.C + c, v[u]);
                if (STAGE == 2 && res) { // the residual loads travel with the tensor's (one latency per batch, not one per pixel)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        rp[u] = res_pixel(p + static_cast<size_t>(u) * PL);
#pragma unroll
                        for (int k = 0; k < CV; ++k) rv[u][k] = 0.0f;
                        if (rp[u] >= 0) ldv<T, CV>(resn + static_cast<size_t>(rp[u]) * d.C, rv[u]);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) rp[u] = -1;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) consume(v[u], p + static_cast<size_t>(u) * PL, rv[u], rp[u]); // same pixel order as the plain loop
            }
            for (; p < p1; p += PL) {
                float v[CV], rv[CV];
                long rp = -1;
                ldv<T, CV>(xn + p * d.C + c, v);
#pragma unroll
                for (int k = 0; k < CV; ++k) rv[k] = 0.0f;
                if (STAGE == 2 && res) {
                    rp = res_pixel(p);
                    if (rp >= 0) ldv<T, CV>(resn + static_cast<size_t>(rp) * d.C, rv);
                }
                consume(v, p, rv, rp);
            }
        }
        if (STAGE == 0) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < CV; ++k) {
                red[tid * CV + k] = s1[k];
                red[256 * CV + tid * CV + k] = s2[k];
            }
            __syncthreads();
            if (tid < CL * CV) { // thread -> (channel lane tid / CV, component tid % CV): fold the pixel lanes in a fixed order
                const int lane = tid / CV, k = tid % CV;
                float a1 = 0.0f, a2 = 0.0f;
                for (int j = 0; j < PL; ++j) {
                    a1 += red[(j * CL + lane) * CV + k];
                    a2 += red[256 * CV + (j * CL + lane) * CV + k];
                }
                const int cc = c0 + lane * CV + k;
                if (cc < d.C) {
                    float* po = partOut + (static_cast<size_t>(n) * S + s) * 2 * d.C;
                    po[cc] = a1;
                    po[d.C + cc] = a2;
                }
            }
        }
    }
}

// mean[n][c] = p + S1/HW,  mul[n][c] = gamma[c] / sqrt(S2/HW - (S1/HW)^2 + eps)   with S1, S2 summed over the slabs in a fixed order:
// one block per (image, channel), thread t adds slabs t, t+256, ..., then a shared-memory tree (deterministic; a single thread walking
// ~1000 slab partials with dependent L2 loads took longer than both sweeps together)
template <typename T>
__global__ __launch_bounds__(256) void instancenorm_fold_kernel(int NC, int C, int S, int HW, float invHW, float eps, const T* __restrict__ x,
                                                               const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ shift, float* __restrict__ mul) {
    __shared__ float r1[256], r2[256];
    const int i = blockIdx.x;
    const int n = i / C, c = i % C;
    float a1 = 0.0f, a2 = 0.0f;
    for (int j = threadIdx.x; j < S; j += 256) {
        const float* po = part + (static_cast<size_t>(n) * S + j) * 2 * C;
        a1 += po[c];
        a2 += po[C + c];
    }
    r1[threadIdx.x] = a1;
    r2[threadIdx.x] = a2;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (static_cast<int>(threadIdx.x) < w) {
            r1[threadIdx.x] += r1[threadIdx.x + w];
            r2[threadIdx.x] += r2[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    a1 = r1[0];
    a2 = r2[0];
    const float piv = static_cast<float>(x[static_cast<size_t>(n) * HW * C + c]);
    const float m1 = a1 * invHW;
    float var = a2 * invHW - m1 * m1;
    var = var > 0.0f ? var : 0.0f;
    // the normalisation as ONE multiply-add per value, y = x * mul + shift (every consumer -- the normalise sweep below and the convolutions that
    // normalise while they stage, graph rule I -- evaluates exactly this fma, so they stay bit-identical to one another)
    const float mu = gamma[c] / sqrtf(var + eps);
    mul[i] = mu;
    shift[i] = beta[c] - (piv + m1) * mu;
}

// Chain rule F: the producing convolution already reduced every output tile to (mean_t, M2_t) per channel (conv2d_mfma_kernel's LDS
// epilogue), so the statistics sweep over the tensor is not needed.  The tile records are combined with the parallel-variance update
//     n = na + nb,  d = mb - ma,  m = ma + d nb/n,  M2 = M2a + M2b + d^2 na nb/n
// in a fixed order, in two levels so that the reads stay channel-contiguous and enough blocks are in flight (a 1378x818 image is ~9000
// tiles; one block per (image, channel) walking them with a 2*C-float stride took longer than the sweep it replaced):
//   level 1: block = (64-tile chunk, image); thread = (channel lane, 1 of 4 tile lanes) folds 16 tiles, the 4 lanes are folded through LDS
//   level 2: block = image, thread = (channel lane, 1 of 4 chunk lanes): folds the chunk records, writes mean and gamma / sqrt(var + eps)
struct RunStat {
    float n, m, q;
};
__device__ __forceinline__ void stat_merge(RunStat& a, float nb, float mb, float qb) {
    if (nb <= 0.0f) return;
    const float n = a.n + nb, dm = mb - a.m, f = nb / n;
    a.m += dm * f;
    a.q += qb + dm * dm * a.n * f;
    a.n = n;
}

constexpr int kFoldChunk = 64; // tiles per level-1 block

__global__ __launch_bounds__(256) void instancenorm_fold_tiles1_kernel(int C, int H, int W, int tilesX, int tilesY, int TH, int TW, const float* __restrict__ part,
                                                                      float* __restrict__ part2) {
    __shared__ float red[3][256];
    const int n = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    const int tiles = tilesX * tilesY;
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
    const float* pn = part + static_cast<size_t>(n) * tiles * 2 * C;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + cl;
        RunStat a{0.0f, 0.0f, 0.0f};
        if (c < C) {
            const int t0 = chunk * kFoldChunk + tl * (kFoldChunk / 4);
#pragma unroll 4
            for (int j = 0; j < kFoldChunk / 4; ++j) {
                const int t = t0 + j;
                if (t < tiles) {
                    const int ty = t / tilesX, tx = t - ty * tilesX;
                    const float cnt = static_cast<float>(max(0, min(TH, H - ty * TH)) * max(0, min(TW, W - tx * TW)));
                    stat_merge(a, cnt, pn[static_cast<size_t>(t) * 2 * C + c], pn[static_cast<size_t>(t) * 2 * C + C + c]);
                }
            }
        }
        __syncthreads();
        red[0][threadIdx.x] = a.n;
        red[1][threadIdx.x] = a.m;
        red[2][threadIdx.x] = a.q;
        __syncthreads();
        if (tl == 0 && c < C) {
#pragma unroll
            for (int j = 1; j < 4; ++j) stat_merge(a, red[0][j * 64 + cl], red[1][j * 64 + cl], red[2][j * 64 + cl]);
            float* po = part2 + (static_cast<size_t>(n) * chunks + chunk) * 3 * C;
            po[c] = a.n;
            po[C + c] = a.m;
            po[2 * C + c] = a.q;
        }
    }
}

__global__ __launch_bounds__(256) void instancenorm_fold_tiles2_kernel(int C, int chunks, float eps, const float* __restrict__ part2,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ shift,
                                                                      float* __restrict__ mul) {
    __shared__ float red[3][256];
    const int n = blockIdx.x;
    const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6; // channel lane, 1 of 4 chunk lanes (chunks kl, kl + 4, ...)
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + cl;
        RunStat a{0.0f, 0.0f, 0.0f};
        if (c < C)
            for (int k = kl; k < chunks; k += 4) {
                const float* po = part2 + (static_cast<size_t>(n) * chunks + k) * 3 * C;
                stat_merge(a, po[c], po[C + c], po[2 * C + c]);
            }
        __syncthreads();
        red[0][threadIdx.x] = a.n;
        red[1][threadIdx.x] = a.m;
        red[2][threadIdx.x] = a.q;
        __syncthreads();
        if (kl == 0 && c < C) {
#pragma unroll
            for (int j = 1; j < 4; ++j) stat_merge(a, red[0][j * 64 + cl], red[1][j * 64 + cl], red[2][j * 64 + cl]);
            float var = a.q / a.n;
            var = var > 0.0f ? var : 0.0f;
            const float mu = gamma[c] / sqrtf(var + eps);
            mul[n * C + c] = mu;
            shift[n * C + c] = beta[c] - a.m * mu; // y = x * mul + shift (see instancenorm_fold_kernel)
        }
    }
}

// ------------------------------------------------------------------------------------------------ plans

struct EltwisePlan : EltwisePlanBase {
    float* d_tab = nullptr;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        const int want = mode == 0 ? 2 : 1;
        SNNHIP_REQUIRE(nIn == want, "%s: expects %d input(s), got %d", desc.c_str(), want, nIn);
        SNNHIP_SAME_DTYPE(desc.c_str());
        const bool v4 = (d.C & 3) == 0;
        if (mode == 0 && !(dims_match(in[0], d.N, d.H, d.W, d.C) && dims_match(in[1], d.N, d.H, d.W, d.C))) {
            // inputs of different extent: output = the plan's dims = max over the inputs (see add_ragged_kernel)
            for (int i = 0; i < 2; ++i)
                SNNHIP_REQUIRE(in[i]->n == d.N && in[i]->c == d.C && in[i]->h <= d.H && in[i]->w <= d.W, "%s: input %d dims %dx%dx%dx%d do not fit %dx%dx%dx%d",
                               desc.c_str(), i, in[i]->n, in[i]->h, in[i]->w, in[i]->c, d.N, d.H, d.W, d.C);
            SNNHIP_REQUIRE(dims_match(out, d.N, d.H, d.W, d.C), "%s: output dims mismatch", desc.c_str());
            const unsigned gg = grid_for(ctx, out->count() / (v4 ? 4 : 1));
            SNNHIP_WITH_T(out->dtype,
                if (v4) SNNHIP_LAUNCH((add_ragged_kernel<4, T>), dim3(gg), dim3(256), 0, ctx->stream, d.N, d.H, d.W, d.C, in[0]->h, in[0]->w, in[1]->h,
                                           in[1]->w, d.act, d.leaky, cptr<T>(in[0]), cptr<T>(in[1]), mptr<T>(out));
                else SNNHIP_LAUNCH((add_ragged_kernel<1, T>), dim3(gg), dim3(256), 0, ctx->stream, d.N, d.H, d.W, d.C, in[0]->h, in[0]->w, in[1]->h,
                                        in[1]->w, d.act, d.leaky, cptr<T>(in[0]), cptr<T>(in[1]), mptr<T>(out)););
            SNNHIP_CHECK_HIP(hipGetLastError());
            return SNNHIP_OK;
        }
        for (int i = 0; i < nIn; ++i)
            SNNHIP_REQUIRE(dims_match(in[i], d.N, d.H, d.W, d.C), "%s: input %d dims %dx%dx%dx%d != plan %dx%dx%dx%d", desc.c_str(), i, in[i]->n, in[i]->h,
                           in[i]->w, in[i]->c, d.N, d.H, d.W, d.C);
        // the output may be any reshape of the same batch (Flatten writes [N,1,1,H*W*C] / [N,1,H*W*C,1]; NHWC memory is the HWC flatten order)
        SNNHIP_REQUIRE(out->n == d.N && out->count() == in[0]->count(), "%s: output %dx%dx%dx%d is not a reshape of the input", desc.c_str(), out->n,
                       out->h, out->w, out->c);
        const size_t count = out->count();
        const unsigned g = grid_for(ctx, v4 ? count / 4 : count);
        const float4* tab = reinterpret_cast<const float4*>(d_tab);
#define SNNHIP_ELT(M)                                                                                                                            \
    SNNHIP_WITH_T(out->dtype, const T* bb = M == 0 ? cptr<T>(in[1]) : nullptr;                                                                   \
                  if (v4) SNNHIP_LAUNCH((eltwise_kernel<M, 4, T>), dim3(g), dim3(256), 0, ctx->stream, count, d.C, d.act, d.leaky, cptr<T>(in[0]), bb, \
                                             tab, mptr<T>(out));                                                                                 \
                  else SNNHIP_LAUNCH((eltwise_kernel<M, 1, T>), dim3(g), dim3(256), 0, ctx->stream, count, d.C, d.act, d.leaky, cptr<T>(in[0]), bb, tab, \
                                          mptr<T>(out));)
        if (mode == 0) {
            SNNHIP_ELT(0);
        } else if (mode == 1) {
            SNNHIP_ELT(1);
        } else {
            SNNHIP_ELT(2);
        }
#undef SNNHIP_ELT
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

int make_eltwise(snnhip_ctx* ctx, const snnhip_eltwise_desc* d, int mode, const std::vector<float>* tab, const char* name, snnhip_plan** out) {
    auto* plan = new EltwisePlan();
    plan->ctx = ctx;
    plan->anyDtype = true;
    plan->d = *d;
    plan->mode = mode;
    plan->numInputs = mode == 0 ? 2 : 1;
    if (tab) {
        int rc = plan->upload(tab->data(), tab->size(), &plan->d_tab);
        if (rc != SNNHIP_OK) {
            delete plan;
            return rc;
        }
    }
    const double cnt = static_cast<double>(d->N) * d->H * d->W * d->C;
    plan->inDims[0] = plan->outDims[0] = d->N; plan->inDims[1] = plan->outDims[1] = d->H;
    plan->inDims[2] = plan->outDims[2] = d->W; plan->inDims[3] = plan->outDims[3] = d->C;
    plan->flops = cnt * (mode == 2 ? 3 : 1);
    plan->bytes = 4.0 * cnt * (mode == 0 ? 3 : 2);
    char buf[160];
    snprintf(buf, sizeof(buf), "%s %dx%dx%dx%d act=%d%s", name, d->N, d->H, d->W, d->C, d->act, (d->C & 3) ? " scalar" : " vec4");
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

struct PoolPlan : snnhip_plan {
    snnhip_pool2d_desc d;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "pool2d: expects 1 input, got %d", nIn);
        SNNHIP_SAME_DTYPE("pool2d");
        SNNHIP_REQUIRE(dims_match(in[0], d.N, d.H, d.W, d.C), "pool2d: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", in[0]->n, in[0]->h, in[0]->w,
                       in[0]->c, d.N, d.H, d.W, d.C);
        SNNHIP_REQUIRE(dims_match(out, d.N, d.OH, d.OW, d.C), "pool2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n, out->h, out->w, out->c, d.N,
                       d.OH, d.OW, d.C);
        const bool vec = (d.C & 3) == 0;
        if (vec && d.type == SNNHIP_POOL_AVG && d.OH == 1 && d.OW == 1 && d.padT == 0 && d.padL == 0 && d.kh >= d.H && d.kw >= d.W && d.H * d.W >= 16) {
            const size_t outs = static_cast<size_t>(d.N) * (d.C >> 2);
            SNNHIP_WITH_T(out->dtype, SNNHIP_LAUNCH((global_avgpool_kernel<T>), dim3(static_cast<unsigned>((outs + 15) / 16)), dim3(256), 0, ctx->stream, d.N,
                                                         d.H * d.W, d.C, cptr<T>(in[0]), mptr<T>(out)););
            SNNHIP_CHECK_HIP(hipGetLastError());
            return SNNHIP_OK;
        }
        const unsigned g = grid_for(ctx, out->count() / (vec ? 4 : 1));
#define SNNHIP_POOL(TY)                                                                                                                   \
    SNNHIP_WITH_T(out->dtype, if (vec) SNNHIP_LAUNCH((pool2d_kernel<TY, 4, T>), dim3(g), dim3(256), 0, ctx->stream, d, cptr<T>(in[0]), mptr<T>(out)); \
                  else SNNHIP_LAUNCH((pool2d_kernel<TY, 1, T>), dim3(g), dim3(256), 0, ctx->stream, d, cptr<T>(in[0]), mptr<T>(out));)
        if (d.type == SNNHIP_POOL_MAX) {
            SNNHIP_POOL(SNNHIP_POOL_MAX);
        } else {
            SNNHIP_POOL(SNNHIP_POOL_AVG);
        }
#undef SNNHIP_POOL
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

struct PadPlan : PadPlanBase {
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "pad: expects 1 input, got %d", nIn);
        SNNHIP_SAME_DTYPE("pad");
        SNNHIP_REQUIRE(dims_match(in[0], d.N, d.H, d.W, d.C) && dims_match(out, d.N, OH, OW, d.C), "pad: tensor dims do not match the plan");
        const bool vec = (d.C & 3) == 0;
        const unsigned g = grid_for(ctx, out->count() / (vec ? 4 : 1));
        SNNHIP_WITH_T(out->dtype, if (vec) SNNHIP_LAUNCH((pad_kernel<4, T>), dim3(g), dim3(256), 0, ctx->stream, d, OH, OW, cptr<T>(in[0]), mptr<T>(out));
                      else SNNHIP_LAUNCH((pad_kernel<1, T>), dim3(g), dim3(256), 0, ctx->stream, d, OH, OW, cptr<T>(in[0]), mptr<T>(out)););
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

struct UpsamplePlan : UpsamplePlanBase {
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "upsample: expects 1 input, got %d", nIn);
        SNNHIP_SAME_DTYPE("upsample");
        SNNHIP_REQUIRE(dims_match(in[0], d.N, d.H, d.W, d.C) && dims_match(out, d.N, OH, OW, d.C), "upsample: tensor dims do not match the plan");
        const bool vec = (d.C & 3) == 0;
        const unsigned g = grid_for(ctx, out->count() / (vec ? 4 : 1));
        const float inv = 1.0f / d.scale; // upsampling2dVulkan.cpp:101
#define SNNHIP_UP(M)                                                                                                                          \
    SNNHIP_WITH_T(out->dtype,                                                                                                                 \
                  if (vec) SNNHIP_LAUNCH((upsample_kernel<M, 4, T>), dim3(g), dim3(256), 0, ctx->stream, d, OH, OW, inv, cptr<T>(in[0]), mptr<T>(out)); \
                  else SNNHIP_LAUNCH((upsample_kernel<M, 1, T>), dim3(g), dim3(256), 0, ctx->stream, d, OH, OW, inv, cptr<T>(in[0]), mptr<T>(out));)
        if (d.mode == SNNHIP_UPSAMPLE_NEAREST) {
            SNNHIP_UP(SNNHIP_UPSAMPLE_NEAREST);
        } else {
            SNNHIP_UP(SNNHIP_UPSAMPLE_BILINEAR);
        }
#undef SNNHIP_UP
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

struct InstanceNormPlan : snnhip_plan {
    snnhip_instancenorm_desc d;
    int S = 1, pixelsPerSlab = 1, CLs = 2;
    float *d_beta = nullptr, *d_gamma = nullptr, *d_part = nullptr, *d_mean = nullptr /* shift[n][c] = beta - mean * mul */, *d_mul = nullptr;
    float* d_foldScratch = nullptr; // chain rule F: level-1 records of the tile-statistics fold
    size_t foldScratchCount = 0;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "instancenorm: expects 1 input, got %d", nIn);
        return runWithResidual(in, nIn, out, nullptr, 0, 0.0f);
    }
    // res != nullptr: the Add layer behind this norm folded into the normalise sweep (chain rule H)
    // statsOnly (graph rule I): the statistics sweep and the fold only -- d_mean / d_mul are left for the convolution that normalises while it
    // stages its input (`out` is not touched and may be the input itself)
    // tiles (chain rule F): the producing convolution left per-tile statistics -- a fold over those records replaces the statistics sweep
    int runWithResidual(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out, const snnhip_tensor* res, int addAct, float addLeaky, bool resFirst = false,
                        bool statsOnly = false, const TileStatsRef* tiles = nullptr) {
        SNNHIP_SAME_DTYPE("instancenorm");
        SNNHIP_REQUIRE(dims_match(in[0], d.N, d.H, d.W, d.C) && dims_match(out, d.N, d.H, d.W, d.C), "instancenorm: tensor dims do not match the plan");
        SNNHIP_REQUIRE(!res || (res->n == d.N && res->c == d.C && res->h <= d.H && res->w <= d.W && res->dtype == out->dtype),
                       "instancenorm+add: the residual does not fit the output");
        InResidual ra;
        if (res) {
            ra.p = res->data;
            ra.H = res->h;
            ra.W = res->w;
            ra.zeroOutside = resFirst ? 1 : 0;
            ra.act = addAct;
            ra.leaky = addLeaky;
        }
        const dim3 g(static_cast<unsigned>(d.N * S));
        const int NC = d.N * d.C, HW = d.H * d.W;
        const dim3 gf(static_cast<unsigned>(NC)); // one block per (image, channel)
        const float invHW = 1.0f / (static_cast<float>(d.H) * static_cast<float>(d.W));
        const bool fastNorm = (d.act == SNNHIP_ACT_NONE || d.act == SNNHIP_ACT_RELU) && (!res || addAct == SNNHIP_ACT_NONE) && !snnhip::option("SNNHIP_NORM_GENERIC_ACT");
#define SNNHIP_IN(ST, CVV)                                                                                                                                     \
    do {                                                                                                                                                       \
        if (ST == 2 && fastNorm && d.act == SNNHIP_ACT_RELU)                                                                                                   \
            SNNHIP_LAUNCH((instancenorm_kernel<ST, CVV, T, ST == 2 ? 1 : 0>), g, dim3(256), 0, ctx->stream, d, S, pixelsPerSlab, CLs, cptr<T>(in[0]), d_mean, d_mul, \
                          d_beta, d_part, mptr<T>(out), ST == 2 ? ra : InResidual());                                                                          \
        else if (ST == 2 && fastNorm)                                                                                                                          \
            SNNHIP_LAUNCH((instancenorm_kernel<ST, CVV, T, ST == 2 ? 2 : 0>), g, dim3(256), 0, ctx->stream, d, S, pixelsPerSlab, CLs, cptr<T>(in[0]), d_mean, d_mul, \
                          d_beta, d_part, mptr<T>(out), ST == 2 ? ra : InResidual());                                                                          \
        else                                                                                                                                                   \
            SNNHIP_LAUNCH((instancenorm_kernel<ST, CVV, T, 0>), g, dim3(256), 0, ctx->stream, d, S, pixelsPerSlab, CLs, cptr<T>(in[0]), d_mean, d_mul,    \
                          d_beta, d_part, mptr<T>(out), ST == 2 ? ra : InResidual());                                                                          \
    } while (0)
#define SNNHIP_FOLD() \
    SNNHIP_LAUNCH(instancenorm_fold_kernel<T>, gf, dim3(256), 0, ctx->stream, NC, d.C, S, HW, invHW, d.eps, cptr<T>(in[0]), d_part, d_gamma, d_beta, d_mean, d_mul)
        const bool sweep = !(tiles && tiles->part);
        // launch trace: each pass under its own scope with the bytes that pass has to move (the statistics sweep reads the tensor once, the
        // normalise sweep reads it [and the residual] once and writes it once; the folds move a few KB)
        const double tensorBytes = static_cast<double>(d.N) * d.H * d.W * d.C * (out->dtype == SNNHIP_F16 ? 2.0 : 4.0);
        const double resBytes = res ? static_cast<double>(res->n) * res->h * res->w * res->c * (out->dtype == SNNHIP_F16 ? 2.0 : 4.0) : 0.0;
        if (!sweep && !tiles->folded) {
            TraceScope ts(desc + " [fold of tile statistics]", 0.0, 0.0);
            const int rc = foldTiles(*tiles);
            if (rc != SNNHIP_OK) return rc;
        }
        SNNHIP_WITH_T(out->dtype, if ((d.C & 3) == 0) {
            if (sweep) {
                {
                    TraceScope ts(desc + " [statistics sweep]", 2.0 * tensorBytes / (out->dtype == SNNHIP_F16 ? 2.0 : 4.0), tensorBytes);
                    SNNHIP_IN(0, 4);
                }
                TraceScope ts(desc + " [fold]", 0.0, 0.0);
                SNNHIP_FOLD();
            }
            if (!statsOnly) {
                TraceScope ts(desc + " [normalise sweep]", 2.0 * tensorBytes / (out->dtype == SNNHIP_F16 ? 2.0 : 4.0), 2.0 * tensorBytes + resBytes);
                // half tensors with whole 8-channel groups, branch-free form: 16 bytes per access (SNNHIP_NORM_CV8=0: 8).  With the per-value activation
                // switch in the loop the wider access measured 0.73x (round 3); without it the sweep is a plain stream.
                const char* cv8 = snnhip::option("SNNHIP_NORM_CV8");
                if (sizeof(T) == 2 && fastNorm && (d.C & 7) == 0 && CLs >= 1 && !(cv8 && atoi(cv8) == 0)) {
                    const int CLs4 = CLs;
                    {
                        const int CLs = CLs4 - 1; // half as many channel lanes, each twice as wide
                        if (d.act == SNNHIP_ACT_RELU)
                            SNNHIP_LAUNCH((instancenorm_kernel<2, 8, T, 1>), g, dim3(256), 0, ctx->stream, d, S, pixelsPerSlab, CLs, cptr<T>(in[0]), d_mean, d_mul, d_beta,
                                          d_part, mptr<T>(out), ra);
                        else
                            SNNHIP_LAUNCH((instancenorm_kernel<2, 8, T, 2>), g, dim3(256), 0, ctx->stream, d, S, pixelsPerSlab, CLs, cptr<T>(in[0]), d_mean, d_mul, d_beta,
                                          d_part, mptr<T>(out), ra);
                    }
                } else {
                    SNNHIP_IN(2, 4);
                }
            }
        } else {
            if (sweep) {
                {
                    TraceScope ts(desc + " [statistics sweep]", 2.0 * tensorBytes / (out->dtype == SNNHIP_F16 ? 2.0 : 4.0), tensorBytes);
                    SNNHIP_IN(0, 1);
                }
                TraceScope ts(desc + " [fold]", 0.0, 0.0);
                SNNHIP_FOLD();
            }
            if (!statsOnly) {
                TraceScope ts(desc + " [normalise sweep]", 2.0 * tensorBytes / (out->dtype == SNNHIP_F16 ? 2.0 : 4.0), 2.0 * tensorBytes + resBytes);
                SNNHIP_IN(2, 1);
            }
        });
#undef SNNHIP_IN
#undef SNNHIP_FOLD
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
    int foldTiles(const TileStatsRef& t); // below the fold kernels' launch geometry
};

// chain rule H: InstanceNorm -> Add(., residual) as the norm's own three launches; borrows the norm plan (parameters and scratch)
struct InstanceNormAddPlan : snnhip_plan {
    InstanceNormPlan* norm = nullptr;
    int addAct = 0;
    float addLeaky = 0.0f;
    bool resFirst = false; // the residual is the Add's first input (matters only when it is smaller than the norm)
    TileStatsRef tiles;    // chain rule F: statistics from the producing convolution's tile records instead of a sweep
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 2, "instancenorm+add: expects 2 inputs (x, residual), got %d", nIn);
        return norm->runWithResidual(in, 1, out, in[1], addAct, addLeaky, resFirst, false, &tiles);
    }
};

} // namespace

int make_instancenorm_add_plan(snnhip_ctx* ctx, snnhip_plan* normPlan, snnhip_plan* addPlan, bool normIsFirstInput, snnhip_plan** out) {
    auto* q = dynamic_cast<InstanceNormPlan*>(normPlan);
    auto* ad = dynamic_cast<EltwisePlanBase*>(addPlan);
    if (!q || !ad || ad->mode != 0 || snnhip::option("SNNHIP_NO_ADD_FUSION")) return SNNHIP_E_UNSUPPORTED;
    if (ad->d.N != q->d.N || ad->d.H != q->d.H || ad->d.W != q->d.W || ad->d.C != q->d.C) return SNNHIP_E_UNSUPPORTED;
    auto* plan = new InstanceNormAddPlan();
    plan->ctx = ctx;
    plan->norm = q;
    plan->addAct = ad->d.act;
    plan->addLeaky = ad->d.leaky;
    plan->resFirst = !normIsFirstInput;
    plan->anyDtype = true;
    plan->numInputs = 2;
    memcpy(plan->inDims, q->inDims, sizeof(plan->inDims));
    memcpy(plan->outDims, q->outDims, sizeof(plan->outDims));
    plan->flops = q->flops + ad->flops;
    plan->bytes = q->bytes + ad->bytes;
    plan->desc = q->desc + " +add act=" + std::to_string(ad->d.act) + (plan->resFirst ? " (residual first)" : "");
    *out = plan;
    return SNNHIP_OK;
}

// graph rule I: where the norm's per-(image, channel) multiplier and shift (y = x * mul + shift) live (device pointers, stable for the life of the plan), and the
// two launches that fill them from a tensor
bool instancenorm_stat_pointers(const snnhip_plan* plan, const float** shift, const float** mul) {
    const auto* q = dynamic_cast<const InstanceNormPlan*>(plan);
    if (!q) return false;
    *shift = q->d_mean;
    *mul = q->d_mul;
    return true;
}
int instancenorm_run_stats(snnhip_plan* plan, const snnhip_tensor* x, const TileStatsRef* tiles) {
    auto* q = dynamic_cast<InstanceNormPlan*>(plan);
    SNNHIP_REQUIRE(q && x, "instancenorm_run_stats: bad arguments");
    snnhip_tensor alias = *x; // stage 2 does not run: the output is never written
    const snnhip_tensor* ins[1] = {x};
    return q->runWithResidual(ins, 1, &alias, nullptr, 0, 0.0f, false, true, tiles);
}

bool instancenorm_fold_target(snnhip_plan* normPlan, NormFoldTarget* t) {
    auto* q = dynamic_cast<InstanceNormPlan*>(normPlan);
    if (!q) return false;
    t->gamma = q->d_gamma;
    t->beta = q->d_beta;
    t->shift = q->d_mean;
    t->mul = q->d_mul;
    t->eps = q->d.eps;
    return true;
}

snnhip_plan* instancenorm_add_use_tile_stats(snnhip_plan* plan, const TileStatsRef& tiles) {
    auto* q = dynamic_cast<InstanceNormAddPlan*>(plan);
    if (!q) return nullptr;
    q->tiles = tiles;
    if (tiles.part)
        q->desc = "instancenorm " + std::to_string(q->norm->d.N) + "x" + std::to_string(q->norm->d.H) + "x" + std::to_string(q->norm->d.W) + "x" + std::to_string(q->norm->d.C) +
              " act=" + std::to_string(q->norm->d.act) + (tiles.folded ? " (statistics from the convolution, 1 sweep) +add act=" : " (fold of tile stats + 1 sweep) +add act=") + std::to_string(q->addAct) + (q->resFirst ? " (residual first)" : "");
    return q->norm;
}

bool pool2d_plan_desc(const snnhip_plan* plan, snnhip_pool2d_desc* d) {
    const auto* pp = dynamic_cast<const PoolPlan*>(plan);
    if (!pp) return false;
    if (d) *d = pp->d;
    return true;
}

bool instancenorm_plan_desc(const snnhip_plan* plan, snnhip_instancenorm_desc* d) {
    const auto* q = dynamic_cast<const InstanceNormPlan*>(plan);
    if (!q) return false;
    if (d) *d = q->d;
    return true;
}

// sizes the level-1 fold records for a convolution tile grid of tilesX x tilesY (chain rule F calls it when the chain is created)
int instancenorm_reserve_tile_stats(snnhip_plan* inPlan, int tilesX, int tilesY) {
    auto* q = dynamic_cast<InstanceNormPlan*>(inPlan);
    SNNHIP_REQUIRE(q && tilesX > 0 && tilesY > 0, "instancenorm_reserve_tile_stats: bad arguments");
    const int tiles = tilesX * tilesY, chunks = (tiles + kFoldChunk - 1) / kFoldChunk;
    const size_t need = static_cast<size_t>(q->d.N) * chunks * 3 * q->d.C;
    if (q->foldScratchCount < need) {
        void* buf = nullptr;
        SNNHIP_CHECK_HIP(snnhip::dev_malloc(&buf, need * sizeof(float)));
        q->deviceAllocs.push_back(buf);
        q->d_foldScratch = static_cast<float*>(buf);
        q->foldScratchCount = need;
    }
    return SNNHIP_OK;
}

int InstanceNormPlan::foldTiles(const TileStatsRef& t) {
    const int tiles = t.tilesX * t.tilesY, chunks = (tiles + kFoldChunk - 1) / kFoldChunk;
    const size_t need = static_cast<size_t>(d.N) * chunks * 3 * d.C;
    SNNHIP_REQUIRE(foldScratchCount >= need, "instancenorm: fold scratch not reserved for a %d x %d tile grid (instancenorm_reserve_tile_stats)", t.tilesX, t.tilesY);
    SNNHIP_LAUNCH(instancenorm_fold_tiles1_kernel, dim3(static_cast<unsigned>(chunks), static_cast<unsigned>(d.N)), dim3(256), 0, ctx->stream, d.C, d.H, d.W,
                       t.tilesX, t.tilesY, t.TH, t.TW, t.part, d_foldScratch);
    SNNHIP_LAUNCH(instancenorm_fold_tiles2_kernel, dim3(static_cast<unsigned>(d.N)), dim3(256), 0, ctx->stream, d.C, chunks, d.eps, d_foldScratch, d_gamma, d_beta,
                       d_mean, d_mul);
    SNNHIP_CHECK_HIP(hipGetLastError());
    return SNNHIP_OK;
}

int instancenorm_apply_tile_stats(snnhip_plan* inPlan, const TileStatsRef& tiles, snnhip_tensor* xy) {
    auto* q = dynamic_cast<InstanceNormPlan*>(inPlan);
    SNNHIP_REQUIRE(q && tiles.part && xy, "instancenorm_apply_tile_stats: bad arguments");
    const snnhip_tensor* ins[1] = {xy};
    return q->runWithResidual(ins, 1, xy, nullptr, 0, 0.0f, false, false, &tiles); // fold, then the normalise sweep in place
}

} // namespace snnhip

using namespace snnhip;

static int check_eltwise(const snnhip_ctx* ctx, const snnhip_eltwise_desc* d, snnhip_plan** out, const char* what) {
    SNNHIP_REQUIRE(ctx && d && out, "%s: null argument", what);
    SNNHIP_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0, "%s: bad dims %dx%dx%dx%d", what, d->N, d->H, d->W, d->C);
    SNNHIP_REQUIRE(d->act >= SNNHIP_ACT_NONE && d->act <= SNNHIP_ACT_SILU, "%s: activation id %d", what, d->act);
    return SNNHIP_OK;
}

extern "C" {

int snnhip_add_plan_create(snnhip_ctx* ctx, const snnhip_eltwise_desc* desc, snnhip_plan** out) {
    int rc = check_eltwise(ctx, desc, out, "add_plan_create");
    return rc != SNNHIP_OK ? rc : make_eltwise(ctx, desc, 0, nullptr, "add", out);
}

int snnhip_activation_plan_create(snnhip_ctx* ctx, const snnhip_eltwise_desc* desc, snnhip_plan** out) {
    int rc = check_eltwise(ctx, desc, out, "activation_plan_create");
    return rc != SNNHIP_OK ? rc : make_eltwise(ctx, desc, 1, nullptr, "activation", out);
}

int snnhip_batchnorm_plan_create(snnhip_ctx* ctx, const snnhip_eltwise_desc* desc, const float* beta, const float* gamma, const float* mean,
                                 const float* var, snnhip_plan** out) {
    int rc = check_eltwise(ctx, desc, out, "batchnorm_plan_create");
    if (rc != SNNHIP_OK) return rc;
    SNNHIP_REQUIRE(beta && gamma && mean && var, "batchnorm_plan_create: null parameter array");
    std::vector<float> tab(static_cast<size_t>(desc->C) * 4, 0.0f);
    for (int c = 0; c < desc->C; ++c) {
        float sq = sqrtf(var[c] + 0.001f);
        sq = sq > 0.0001f ? sq : 0.0001f; // vk_batchnorm.comp:66-67
        tab[4 * c + 0] = gamma[c] / sq;
        tab[4 * c + 1] = mean[c];
        tab[4 * c + 2] = beta[c];
    }
    return make_eltwise(ctx, desc, 2, &tab, "batchnorm", out);
}

int snnhip_pool2d_plan_create(snnhip_ctx* ctx, const snnhip_pool2d_desc* desc, snnhip_plan** out) {
    SNNHIP_REQUIRE(ctx && desc && out, "pool2d_plan_create: null argument");
    SNNHIP_REQUIRE(desc->N > 0 && desc->H > 0 && desc->W > 0 && desc->C > 0 && desc->kh > 0 && desc->kw > 0 && desc->sh > 0 && desc->sw > 0,
                   "pool2d desc: bad dims");
    SNNHIP_REQUIRE(desc->type == SNNHIP_POOL_MAX || desc->type == SNNHIP_POOL_AVG, "pool2d desc: type %d", desc->type);
    SNNHIP_REQUIRE(desc->padT >= 0 && desc->padL >= 0, "pool2d desc: negative padding");
    auto* plan = new PoolPlan();
    plan->ctx = ctx;
    plan->anyDtype = true;
    plan->d = *desc;
    auto outdim = [&](int in, int k, int s) { // maxpool2d.cpp:26-36 / avgpool2d.cpp:20-29 through genericlayer.cpp:64-90 (max(0, translation))
        const float scale = 1.0f / static_cast<float>(s);
        float tr = desc->same ? 1.0f - 1.0f / static_cast<float>(s) : 1.0f - static_cast<float>(k) / static_cast<float>(s);
        if (tr < 0.0f) tr = 0.0f;
        return static_cast<int>(static_cast<unsigned>(scale * static_cast<float>(in) + tr));
    };
    if (plan->d.OH <= 0) plan->d.OH = outdim(desc->H, desc->kh, desc->sh);
    if (plan->d.OW <= 0) plan->d.OW = outdim(desc->W, desc->kw, desc->sw);
    if (plan->d.OH <= 0 || plan->d.OW <= 0) {
        set_error("pool2d desc: empty output %dx%d", plan->d.OH, plan->d.OW);
        delete plan;
        return SNNHIP_E_INVALID;
    }
    const snnhip_pool2d_desc& d = plan->d;
    plan->inDims[0] = d.N; plan->inDims[1] = d.H; plan->inDims[2] = d.W; plan->inDims[3] = d.C;
    plan->outDims[0] = d.N; plan->outDims[1] = d.OH; plan->outDims[2] = d.OW; plan->outDims[3] = d.C;
    plan->flops = static_cast<double>(d.N) * d.OH * d.OW * d.C * d.kh * d.kw;
    plan->bytes = 4.0 * (static_cast<double>(d.N) * d.H * d.W * d.C + static_cast<double>(d.N) * d.OH * d.OW * d.C);
    char buf[160];
    snprintf(buf, sizeof(buf), "%spool2d k=%dx%d s=%d c=%d %dx%d->%dx%d%s", d.type == SNNHIP_POOL_MAX ? "max" : "avg", d.kh, d.kw, d.sh, d.C, d.H, d.W,
             d.OH, d.OW, (d.C & 3) ? " scalar" : " vec4");
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

int snnhip_pad_plan_create(snnhip_ctx* ctx, const snnhip_pad_desc* desc, snnhip_plan** out) {
    SNNHIP_REQUIRE(ctx && desc && out, "pad_plan_create: null argument");
    SNNHIP_REQUIRE(desc->N > 0 && desc->H > 0 && desc->W > 0 && desc->C > 0, "pad desc: bad dims");
    SNNHIP_REQUIRE(desc->padT >= 0 && desc->padB >= 0 && desc->padL >= 0 && desc->padR >= 0, "pad desc: negative padding");
    SNNHIP_REQUIRE(desc->mode >= 0 && desc->mode <= 2, "pad desc: mode %d", desc->mode);
    auto* plan = new PadPlan();
    plan->ctx = ctx;
    plan->anyDtype = true;
    plan->d = *desc;
    plan->OH = desc->H + desc->padT + desc->padB;
    plan->OW = desc->W + desc->padL + desc->padR;
    plan->inDims[0] = desc->N; plan->inDims[1] = desc->H; plan->inDims[2] = desc->W; plan->inDims[3] = desc->C;
    plan->outDims[0] = desc->N; plan->outDims[1] = plan->OH; plan->outDims[2] = plan->OW; plan->outDims[3] = desc->C;
    plan->bytes = 4.0 * desc->N * desc->C * (static_cast<double>(desc->H) * desc->W + static_cast<double>(plan->OH) * plan->OW);
    char buf[160];
    snprintf(buf, sizeof(buf), "pad mode=%d t%d b%d l%d r%d c=%d %dx%d->%dx%d", desc->mode, desc->padT, desc->padB, desc->padL, desc->padR, desc->C,
             desc->H, desc->W, plan->OH, plan->OW);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

int snnhip_upsample_plan_create(snnhip_ctx* ctx, const snnhip_upsample_desc* desc, snnhip_plan** out) {
    SNNHIP_REQUIRE(ctx && desc && out, "upsample_plan_create: null argument");
    SNNHIP_REQUIRE(desc->N > 0 && desc->H > 0 && desc->W > 0 && desc->C > 0 && desc->scale > 0.0f, "upsample desc: bad dims / scale");
    SNNHIP_REQUIRE(desc->mode == SNNHIP_UPSAMPLE_NEAREST || desc->mode == SNNHIP_UPSAMPLE_BILINEAR, "upsample desc: mode %d", desc->mode);
    auto* plan = new UpsamplePlan();
    plan->ctx = ctx;
    plan->anyDtype = true;
    plan->d = *desc;
    plan->OH = static_cast<int>(static_cast<unsigned>(desc->scale * static_cast<float>(desc->H))); // genericlayer.cpp:76-77, translation 0
    plan->OW = static_cast<int>(static_cast<unsigned>(desc->scale * static_cast<float>(desc->W)));
    if (plan->OH <= 0 || plan->OW <= 0) {
        set_error("upsample desc: empty output");
        delete plan;
        return SNNHIP_E_INVALID;
    }
    plan->inDims[0] = desc->N; plan->inDims[1] = desc->H; plan->inDims[2] = desc->W; plan->inDims[3] = desc->C;
    plan->outDims[0] = desc->N; plan->outDims[1] = plan->OH; plan->outDims[2] = plan->OW; plan->outDims[3] = desc->C;
    plan->bytes = 4.0 * desc->N * desc->C * (static_cast<double>(desc->H) * desc->W + static_cast<double>(plan->OH) * plan->OW);
    char buf[160];
    snprintf(buf, sizeof(buf), "upsample_%s x%g c=%d %dx%d->%dx%d", desc->mode ? "bilinear" : "nearest", desc->scale, desc->C, desc->H, desc->W, plan->OH,
             plan->OW);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

int snnhip_instancenorm_plan_create(snnhip_ctx* ctx, const snnhip_instancenorm_desc* desc, const float* beta, const float* gamma, snnhip_plan** out) {
    SNNHIP_REQUIRE(ctx && desc && out && beta && gamma, "instancenorm_plan_create: null argument");
    SNNHIP_REQUIRE(desc->N > 0 && desc->H > 0 && desc->W > 0 && desc->C > 0, "instancenorm desc: bad dims");
    SNNHIP_REQUIRE(desc->act >= SNNHIP_ACT_NONE && desc->act <= SNNHIP_ACT_SILU, "instancenorm desc: activation id %d", desc->act);
    auto* plan = new InstanceNormPlan();
    plan->ctx = ctx;
    plan->anyDtype = true;
    plan->d = *desc;
    // slabs = pixel ranges of an image: enough blocks to cover the chip about 8 times (HBM-bound sweeps want many waves in flight), at least
    // 4 pixels per pixel lane each (the unrolled loop) and a whole number of 64-pixel groups so that slab starts stay vector-aligned
    const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
    const long long HWp = static_cast<long long>(desc->H) * desc->W;
    long long S = (8LL * cus + desc->N - 1) / desc->N;
    long long pps = (HWp + S - 1) / S;
    if (pps < 256) pps = 256;
    pps = (pps + 63) / 64 * 64;
    plan->pixelsPerSlab = static_cast<int>(pps);
    plan->S = static_cast<int>((HWp + pps - 1) / pps);
    int rc = plan->upload(beta, desc->C, &plan->d_beta);
    if (rc == SNNHIP_OK) rc = plan->upload(gamma, desc->C, &plan->d_gamma);
    {   // channel lanes: enough to read whole pixels (up to 128 channels = 512 contiguous bytes) per wave instruction
        const int cv = (desc->C & 3) == 0 ? 4 : 1;
        int cls = 2;
        while ((1 << cls) * cv < desc->C && cls < (cv == 4 ? 5 : 6)) ++cls;
        plan->CLs = cls;
    }
    std::vector<float> zeros(static_cast<size_t>(desc->N) * plan->S * desc->C * 2, 0.0f);
    if (rc == SNNHIP_OK) rc = plan->upload(zeros.data(), zeros.size(), &plan->d_part);
    if (rc == SNNHIP_OK) rc = plan->upload(zeros.data(), static_cast<size_t>(desc->N) * desc->C, &plan->d_mean);
    if (rc == SNNHIP_OK) rc = plan->upload(zeros.data(), static_cast<size_t>(desc->N) * desc->C, &plan->d_mul);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    const double cnt = static_cast<double>(desc->N) * desc->H * desc->W * desc->C;
    plan->inDims[0] = plan->outDims[0] = desc->N; plan->inDims[1] = plan->outDims[1] = desc->H;
    plan->inDims[2] = plan->outDims[2] = desc->W; plan->inDims[3] = plan->outDims[3] = desc->C;
    plan->flops = cnt * 7;
    plan->bytes = 4.0 * cnt * 2; // algorithmic: read once, write once (the statistics sweep reads it once more: 3x in practice)
    char buf[160];
    snprintf(buf, sizeof(buf), "instancenorm %dx%dx%dx%d act=%d slabs=%d (2 sweeps + fold)", desc->N, desc->H, desc->W, desc->C, desc->act, plan->S);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

} // extern "C"
