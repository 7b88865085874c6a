// depthwise.hip -- NHWC fp32 depthwise convolution (the reference's "SeparableConv2D"/"DepthwiseConv2D" layer).
//
// Replaces shadertemplate_vk_depthwise.comp:64-137: taps that fall outside the image are skipped (zero padding
// by clipping, :77-78), bias buffer always added (:79), fused BN (:91-99) and activation (:101-134).
// HBM-bound: each thread produces 1 pixel x 4 channels with 16-byte channel-contiguous loads/stores; the k*k input
// re-reads of neighbouring pixels are served by L1/L2 (a row of C<=960 channels is <= 3.8 KB per pixel).
#include "epilogue.h"
#include "snnhip_internal.h"

namespace snnhip {
namespace {

struct DwParams {
    int N, H, W, C, kh, kw, sh, sw, padx, pady, act, useBN, OH, OW;
    float leaky;
    int C4; // ceil(C/4)
};

template <bool VEC, typename T>
__global__ __launch_bounds__(256) void depthwise_kernel(DwParams p, const T* __restrict__ x, const float* __restrict__ wpk,
                                                        const float4* __restrict__ epi, T* __restrict__ y) {
    const size_t total = static_cast<size_t>(p.N) * p.OH * p.OW * p.C4;
    for (size_t idx = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; idx < total; idx += static_cast<size_t>(gridDim.x) * 256) {
        const int cq = static_cast<int>(idx % p.C4);
        size_t px = idx / p.C4;
        const int ox = static_cast<int>(px % p.OW);
        px /= p.OW;
        const int oy = static_cast<int>(px % p.OH);
        const int n = static_cast<int>(px / p.OH);
        const int s0x = ox * p.sw - p.padx, s0y = oy * p.sh - p.pady;
        const int sfx = max(0, -s0x), sfy = max(0, -s0y);
        const int efx = min(p.kw, p.W - s0x), efy = min(p.kh, p.H - s0y);
        const int c0 = cq * 4;
        float acc[4] = {0, 0, 0, 0};
        const T* xn = x + static_cast<size_t>(n) * p.H * p.W * p.C;
        for (int fy = sfy; fy < efy; ++fy) {
            for (int fx = sfx; fx < efx; ++fx) {
                const T* xp = xn + (static_cast<size_t>(s0y + fy) * p.W + (s0x + fx)) * p.C + c0;
                const float4 w = *reinterpret_cast<const float4*>(wpk + (static_cast<size_t>(fy) * p.kw + fx) * p.C4 * 4 + c0);
                const float wv[4] = {w.x, w.y, w.z, w.w};
                if (VEC) {
                    float v[4];
                    ldv<T, 4>(xp, v);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[b] = fmaf(v[b], wv[b], acc[b]);
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (c0 + b < p.C) acc[b] = fmaf(static_cast<float>(xp[b]), wv[b], acc[b]);
                }
            }
        }
        float o[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float4 e = epi[c0 + b];
            float v = epi_affine(acc[b], e, p.useBN);
            o[b] = epi_act(p.act == SNNHIP_ACT_SILU_QUIRK ? SNNHIP_ACT_SILU : p.act, p.leaky, v, v);
        }
        T* yo = y + ((static_cast<size_t>(n) * p.OH + oy) * p.OW + ox) * p.C + c0;
        if (VEC) {
            stv<T, 4>(yo, o);
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (c0 + b < p.C) yo[b] = static_cast<T>(o[b]);
        }
    }
}

// 3x3, C % 4 == 0: one thread = a strip of 4 adjacent output pixels x 4 channels.  The 3 x (3*STRIDE + 3) input window is loaded once
// (18 float4 for stride 1, 27 for stride 2, instead of 36) and the 9 weight quads once per strip instead of once per pixel; the
// activation is the branch-free form when it is one of {none, relu, relu6, leakyRelu} (MobileNetV2: relu6 everywhere).
// IDX = unsigned whenever the strip count fits 31 bits (every real layer): the index decomposition below is four divisions per strip, and as
// 64-bit divisions they were a multi-hundred-instruction prologue in front of 18 loads and 108 FMAs.
// ROWS = 2 (stride 1): the thread also owns the strip of the next output row -- 4 input rows feed both (24 float4 for 8 outputs instead of 36).
template <int STRIDE, bool SIMPLE, typename T, typename IDX, int ROWS>
__global__ __launch_bounds__(256) void depthwise3x3_strip_kernel(DwParams p, ActCfg ac, const T* __restrict__ x, const float* __restrict__ wpk,
                                                                 const float4* __restrict__ epi, T* __restrict__ y) {
    constexpr int COLS = 3 * STRIDE + 3; // input columns feeding 4 outputs
    constexpr int INR = 3 + (ROWS - 1) * STRIDE; // input rows feeding ROWS output rows
    const int strips = (p.OW + 3) >> 2;
    const int rowGroups = (p.OH + ROWS - 1) / ROWS;
    const IDX total = static_cast<IDX>(p.N) * rowGroups * strips * p.C4;
    for (IDX idx = static_cast<IDX>(blockIdx.x) * 256 + threadIdx.x; idx < total; idx += static_cast<IDX>(gridDim.x) * 256) {
        const int cq = static_cast<int>(idx % static_cast<IDX>(p.C4));
        IDX r = idx / static_cast<IDX>(p.C4);
        const int st = static_cast<int>(r % static_cast<IDX>(strips));
        r /= static_cast<IDX>(strips);
        const int oy = static_cast<int>(r % static_cast<IDX>(rowGroups)) * ROWS;
        const int n = static_cast<int>(r / static_cast<IDX>(rowGroups));
        const int c0 = cq * 4, ox0 = st * 4;
        const int ix0 = ox0 * STRIDE - p.padx, iy0 = oy * STRIDE - p.pady;
        const T* xn = x + static_cast<size_t>(n) * p.H * p.W * p.C + c0;
        float4 w[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const float4*>(wpk + static_cast<size_t>(t) * p.C4 * 4 + c0);
        float4 acc[ROWS][4];
#pragma unroll
        for (int q = 0; q < ROWS; ++q)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[q][a] = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool interior = ix0 >= 0 && iy0 >= 0 && ix0 + COLS <= p.W && iy0 + INR <= p.H;
#pragma unroll
        for (int ry = 0; ry < INR; ++ry) {
            float4 v[COLS];
            const int sy = iy0 + ry;
            auto ld4 = [&](const T* ptr) {
                float t[4];
                ldv<T, 4>(ptr, t);
                return make_float4(t[0], t[1], t[2], t[3]);
            };
            if (interior) {
                const T* row = xn + (static_cast<size_t>(sy) * p.W + ix0) * p.C;
#pragma unroll
                for (int c = 0; c < COLS; ++c) v[c] = ld4(row + static_cast<size_t>(c) * p.C);
            } else {
#pragma unroll
                for (int c = 0; c < COLS; ++c) { // taps outside the image are skipped by the shader == zero contribution
                    const int sx = ix0 + c;
                    v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) v[c] = ld4(xn + (static_cast<size_t>(sy) * p.W + sx) * p.C);
                }
            }
#pragma unroll
            for (int q = 0; q < ROWS; ++q) {
                const int fy = ry - q * STRIDE; // this input row is tap row fy of output row q
                if (fy < 0 || fy > 2) continue;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int fx = 0; fx < 3; ++fx) {
                        const float4 vv = v[a * STRIDE + fx], ww = w[fy * 3 + fx];
                        acc[q][a].x = fmaf(vv.x, ww.x, acc[q][a].x);
                        acc[q][a].y = fmaf(vv.y, ww.y, acc[q][a].y);
                        acc[q][a].z = fmaf(vv.z, ww.z, acc[q][a].z);
                        acc[q][a].w = fmaf(vv.w, ww.w, acc[q][a].w);
                    }
            }
        }
        const float4 e0 = epi[c0], e1 = epi[c0 + 1], e2 = epi[c0 + 2], e3 = epi[c0 + 3];
        const int act = ac.act == SNNHIP_ACT_SILU_QUIRK ? SNNHIP_ACT_SILU : ac.act;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            if (oy + q >= p.OH) break;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (ox0 + a >= p.OW) break;
                float4 o;
                o.x = epi_affine(acc[q][a].x, e0, p.useBN);
                o.y = epi_affine(acc[q][a].y, e1, p.useBN);
                o.z = epi_affine(acc[q][a].z, e2, p.useBN);
                o.w = epi_affine(acc[q][a].w, e3, p.useBN);
                if (SIMPLE) {
                    o = make_float4(apply_act<true>(ac, o.x, 0.f), apply_act<true>(ac, o.y, 0.f), apply_act<true>(ac, o.z, 0.f), apply_act<true>(ac, o.w, 0.f));
                } else {
                    o = make_float4(epi_act(act, ac.leaky, o.x, o.x), epi_act(act, ac.leaky, o.y, o.y), epi_act(act, ac.leaky, o.z, o.z), epi_act(act, ac.leaky, o.w, o.w));
                }
                const float ov[4] = {o.x, o.y, o.z, o.w};
                stv<T, 4>(y + ((static_cast<size_t>(n) * p.OH + oy + q) * p.OW + ox0 + a) * p.C + c0, ov);
            }
        }
    }
}

struct DepthwisePlan : ConvPlanBase {
    DwParams p;
    float* d_w = nullptr;
    float* d_epi = nullptr;

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "depthwise: expects 1 input, got %d", nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.H && x->w == p.W && x->c == p.C, "depthwise: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n,
                       x->h, x->w, x->c, p.N, p.H, p.W, p.C);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.C, "depthwise: output dims %dx%dx%dx%d != plan %dx%dx%dx%d",
                       out->n, out->h, out->w, out->c, p.N, p.OH, p.OW, p.C);
        const bool vec = (p.C % 4) == 0;
        const size_t cap = static_cast<size_t>(ctx->props.multiProcessorCount) * 16;
        if (vec && p.kh == 3 && p.kw == 3 && p.sh == p.sw && (p.sh == 1 || p.sh == 2)) {
            // output rows per thread: 2 for stride 1 while that still leaves two full blocks per CU (144 ch @56x56 b32: 36.5 -> 29.9 us fp32; the
            // 14x14 layers lose parallelism instead: fp16 8.5 -> 9.5 us); SNNHIP_DW_ROWS1 pins 1
            const size_t pairs = static_cast<size_t>(p.N) * ((p.OH + 1) / 2) * ((p.OW + 3) / 4) * p.C4;
            const int rows = (p.sh == 1 && !snnhip::option("SNNHIP_DW_ROWS1") && pairs >= static_cast<size_t>(ctx->props.multiProcessorCount) * 512) ? 2 : 1;
            const size_t total4 = static_cast<size_t>(p.N) * ((p.OH + rows - 1) / rows) * ((p.OW + 3) / 4) * p.C4;
            size_t blocks4 = (total4 + 255) / 256;
            if (blocks4 > cap) blocks4 = cap;
            if (blocks4 == 0) return SNNHIP_OK;
            const ActCfg ac = make_act_cfg(p.act, p.leaky);
            const dim3 g4(static_cast<unsigned>(blocks4));
            const float4* e4 = reinterpret_cast<const float4*>(d_epi);
            const bool simple = act_is_simple(p.act);
            const bool small = total4 + static_cast<size_t>(blocks4) * 256 < 0x7fffffffull; // idx + stride never wraps 32 bits
#define SNNHIP_DW_(ST, SI, TT, IX, RW)                                                                                                             \
    SNNHIP_LAUNCH((depthwise3x3_strip_kernel<ST, SI, TT, IX, RW>), g4, dim3(256), 0, ctx->stream, p, ac, reinterpret_cast<const TT*>(x->data), d_w, e4, \
                       reinterpret_cast<TT*>(out->data))
#define SNNHIP_DW(ST, SI, TT)                                \
    do {                                                     \
        if (ST == 1 && rows == 2) {                          \
            if (small) SNNHIP_DW_(1, SI, TT, unsigned, 2);   \
            else SNNHIP_DW_(1, SI, TT, size_t, 2);           \
        } else {                                             \
            if (small) SNNHIP_DW_(ST, SI, TT, unsigned, 1);  \
            else SNNHIP_DW_(ST, SI, TT, size_t, 1);          \
        }                                                    \
    } while (0)
            if (dtype == SNNHIP_F16) {
                if (p.sh == 1) { if (simple) SNNHIP_DW(1, true, _Float16); else SNNHIP_DW(1, false, _Float16); }
                else { if (simple) SNNHIP_DW(2, true, _Float16); else SNNHIP_DW(2, false, _Float16); }
            } else {
                if (p.sh == 1) { if (simple) SNNHIP_DW(1, true, float); else SNNHIP_DW(1, false, float); }
                else { if (simple) SNNHIP_DW(2, true, float); else SNNHIP_DW(2, false, float); }
            }
#undef SNNHIP_DW
#undef SNNHIP_DW_
            SNNHIP_CHECK_HIP(hipGetLastError());
            return SNNHIP_OK;
        }
        const size_t total = static_cast<size_t>(p.N) * p.OH * p.OW * p.C4;
        size_t blocks = (total + 255) / 256;
        if (blocks > cap) blocks = cap;
        if (blocks == 0) return SNNHIP_OK;
        const dim3 gg(static_cast<unsigned>(blocks));
        const float4* e4 = reinterpret_cast<const float4*>(d_epi);
#define SNNHIP_DWG(V, TT) SNNHIP_LAUNCH((depthwise_kernel<V, TT>), gg, dim3(256), 0, ctx->stream, p, reinterpret_cast<const TT*>(x->data), d_w, e4, reinterpret_cast<TT*>(out->data))
        if (dtype == SNNHIP_F16) {
            if (vec) SNNHIP_DWG(true, _Float16); else SNNHIP_DWG(false, _Float16);
        } else {
            if (vec) SNNHIP_DWG(true, float); else SNNHIP_DWG(false, float);
        }
#undef SNNHIP_DWG
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

} // namespace

int make_depthwise_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_chw, const std::vector<float>& epi4, snnhip_plan** out) {
    if (g.preMode) return SNNHIP_E_UNSUPPORTED; // the fused-Pad address path exists in the MFMA kernel only
    if (g.normShift) return SNNHIP_E_UNSUPPORTED; // graph rule I: not in this kernel

    auto* plan = new DepthwisePlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->depthwise = true;
    const int C = g.OC, taps = g.kh * g.kw;
    plan->w_oihw.assign(w_chw, w_chw + static_cast<size_t>(C) * taps);
    plan->epi4 = epi4;
    DwParams& p = plan->p;
    p.N = g.N; p.H = g.H; p.W = g.W; p.C = C; p.kh = g.kh; p.kw = g.kw; p.sh = g.sh; p.sw = g.sw; p.padx = g.padx; p.pady = g.pady;
    p.act = g.act; p.useBN = g.useBN; p.OH = g.OH; p.OW = g.OW; p.leaky = g.leaky; p.C4 = up_div(C, 4);
    // pack [fy][fx][C4*4] -- the same order SeparableConv2DLayer::oihw2hwo4i4 produces (separableconvolution.cpp:88-111)
    std::vector<float> wpk(static_cast<size_t>(taps) * p.C4 * 4, 0.0f);
    for (int c = 0; c < C; ++c)
        for (int t = 0; t < taps; ++t) wpk[static_cast<size_t>(t) * p.C4 * 4 + c] = w_chw[static_cast<size_t>(c) * taps + t];
    int rc = plan->upload(wpk.data(), wpk.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epi4.data(), epi4.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = g.N; plan->inDims[1] = g.H; plan->inDims[2] = g.W; plan->inDims[3] = C;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = C;
    plan->flops = 2.0 * taps * C * static_cast<double>(g.OH) * g.OW * g.N;
    plan->bytes = 4.0 * (static_cast<double>(g.N) * g.H * g.W * C + static_cast<double>(g.N) * g.OH * g.OW * C + static_cast<double>(C) * taps);
    char buf[200];
    const bool strip = (C % 4) == 0 && g.kh == 3 && g.kw == 3 && g.sh == g.sw && (g.sh == 1 || g.sh == 2);
    snprintf(buf, sizeof(buf), "depthwise_%s k=%dx%d s=%d c=%d %s", g.dtype == SNNHIP_F16 ? "f16" : "f32", g.kh, g.kw, g.sh, C, strip ? "vec4 strip4" : ((C % 4) == 0 ? "vec4" : "scalar"));
    plan->desc = buf;
    plan->dtype = g.dtype;
    if (g.dtype == SNNHIP_F16) plan->bytes *= 0.5;
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
