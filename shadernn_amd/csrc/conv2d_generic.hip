// conv2d_generic.hip -- direct NHWC fp32 convolution for every parameter combination the reference's
// Conv2D layer accepts (any k, stride, IC, OC, padding mode, batch), VALU path.
//
// Replaces shadertemplate_vk_conv2d.comp:148-347 and shadertemplate_vk_conv2d_1x1.comp:68-210 of the reference.
// This is the always-correct variant; GEMM-shaped layers are routed to conv2d_mfma.hip and the ESPCN chain to
// espcn_fused.hip by snnhip_conv2d_plan_create / snnhip_chain_plan_create.
//
// Work decomposition (wave64):
//   block  = 256 threads = 32x8 output pixels x 16 output channels
//   thread = 4 adjacent x-pixels x 4 output channels (the same per-thread tile as the reference shader, so the
//            SiLU 4-pixel-group quirk can be reproduced exactly), 16 fp32 accumulators
//   IC is consumed in chunks of ICB channels staged through LDS:
//       in_tile [tileH][ICB][tileWp]   input halo tile, padding resolved while filling (zero/replicate/reflect)
//       w_tile  [kh*kw][ICB][16]       weights of this block's 16 output channels
//   global reads of the tile are channel-fastest (NHWC-coalesced); stores are 16 B per (pixel, oc-quad).
#include "epilogue.h"
#include "snnhip_internal.h"

namespace snnhip {

namespace {

constexpr int TILE_W = 32, TILE_H = 8, OCB = 16;

struct GenericParams {
    int N, H, W, IC, OC, kh, kw, sh, sw, padx, pady, padMode, act, useBN, OH, OW;
    float leaky;
    int ICB;     // channels per LDS chunk
    int tileH;   // (TILE_H-1)*sh + kh
    int tileW;   // (TILE_W-1)*sw + kw
    int tileWp;  // padded row pitch in floats
    int tilesX;  // number of tiles along x
    int ocBlocks;
};

__global__ __launch_bounds__(256) void conv2d_generic_kernel(GenericParams p, const float* __restrict__ x, const float* __restrict__ wpk,
                                                             const float4* __restrict__ epi, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* in_tile = smem;
    float* w_tile = smem + p.tileH * p.ICB * p.tileWp;

    const int tid = threadIdx.x;
    const int ocq = tid & 3;
    const int pxq = tid >> 2;
    const int qx = pxq & 7, qy = pxq >> 3;

    const int tile = blockIdx.x;
    const int tx = tile % p.tilesX, ty = tile / p.tilesX;
    const int ocb = blockIdx.y;
    const int n = blockIdx.z;

    const int ox0 = tx * TILE_W, oy0 = ty * TILE_H;
    const int ix0 = ox0 * p.sw - p.padx, iy0 = oy0 * p.sh - p.pady;
    const float* xn = x + static_cast<size_t>(n) * p.H * p.W * p.IC;
    const int taps = p.kh * p.kw;

    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;

    for (int ic0 = 0; ic0 < p.IC; ic0 += p.ICB) {
        const int icb = min(p.ICB, p.IC - ic0);
        __syncthreads();
        // ---- stage input tile: idx -> (row, col, ic) with ic fastest (coalesced NHWC read)
        const int fillCount = p.tileH * p.tileW * p.ICB;
        for (int idx = tid; idx < fillCount; idx += 256) {
            int ic = idx % p.ICB;
            int rc = idx / p.ICB;
            int c = rc % p.tileW, r = rc / p.tileW;
            float v = 0.0f;
            if (ic < icb) {
                int sy = resolve_coord(iy0 + r, p.H, p.padMode);
                int sx = resolve_coord(ix0 + c, p.W, p.padMode);
                if (sy >= 0 && sx >= 0) v = xn[(static_cast<size_t>(sy) * p.W + sx) * p.IC + ic0 + ic];
            }
            in_tile[(r * p.ICB + ic) * p.tileWp + c] = v;
        }
        // ---- stage weights: wpk[ocb][tap][IC][16]
        const int wCount = taps * p.ICB * OCB;
        for (int idx = tid; idx < wCount; idx += 256) {
            int o = idx & 15;
            int ti = idx >> 4;
            int ic = ti % p.ICB, tap = ti / p.ICB;
            float v = 0.0f;
            if (ic < icb) v = wpk[((static_cast<size_t>(ocb) * taps + tap) * p.IC + ic0 + ic) * OCB + o];
            w_tile[idx] = v;
        }
        __syncthreads();

        for (int fy = 0; fy < p.kh; ++fy) {
            for (int fx = 0; fx < p.kw; ++fx) {
                const float* wrow = w_tile + ((fy * p.kw + fx) * p.ICB) * OCB + ocq * 4;
                const float* xrow = in_tile + ((qy * p.sh + fy) * p.ICB) * p.tileWp + (qx * 4) * p.sw + fx;
                for (int ic = 0; ic < icb; ++ic) {
                    const float4 w = *reinterpret_cast<const float4*>(wrow + ic * OCB);
                    const float* xr = xrow + ic * p.tileWp;
                    float xv[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) xv[a] = xr[a * p.sw];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        acc[a][0] = fmaf(xv[a], w.x, acc[a][0]);
                        acc[a][1] = fmaf(xv[a], w.y, acc[a][1]);
                        acc[a][2] = fmaf(xv[a], w.z, acc[a][2]);
                        acc[a][3] = fmaf(xv[a], w.w, acc[a][3]);
                    }
                }
            }
        }
    }

    // ---- epilogue + store
    const int oy = oy0 + qy;
#ifdef SNNHIP_GUARD_BREAK // the deliberately broken build tools/sanitize.sh guard must catch: an off-by-one row bound, i.e. one output row stored past the end of the last image
    if (oy > p.OH) return;
#else
    if (oy >= p.OH) return;
#endif
    const int oc0 = ocb * OCB + ocq * 4;
    float4 e[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) e[b] = epi[oc0 + b]; // table is padded to a multiple of 16 channels
    float first[4] = {0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int ox = ox0 + qx * 4 + a;
        float o[4];
        const int act = (p.act == SNNHIP_ACT_SILU_QUIRK && a == 0) ? SNNHIP_ACT_SILU : p.act;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float v = epi_affine(acc[a][b], e[b], p.useBN);
            o[b] = epi_act(act, p.leaky, v, first[b]);
        }
        if (a == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b) first[b] = o[b];
        }
        if (ox < p.OW) {
            float* yo = y + ((static_cast<size_t>(n) * p.OH + oy) * p.OW + ox) * p.OC + oc0;
            if ((p.OC & 3) == 0 && oc0 + 3 < p.OC) {
                *reinterpret_cast<float4*>(yo) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (oc0 + b < p.OC) yo[b] = o[b];
            }
        }
    }
}

struct GenericConvPlan : ConvPlanBase {
    GenericParams p;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    size_t ldsBytes = 0;

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "conv2d: expects 1 input, got %d", nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.H && x->w == p.W && x->c == p.IC, "conv2d: input dims %dx%dx%dx%d != plan %dx%dx%dx%d",
                       x->n, x->h, x->w, x->c, p.N, p.H, p.W, p.IC);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.OC, "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d",
                       out->n, out->h, out->w, out->c, p.N, p.OH, p.OW, p.OC);
        dim3 grid(p.tilesX * up_div(p.OH, TILE_H), p.ocBlocks, p.N);
        SNNHIP_LAUNCH(conv2d_generic_kernel, grid, dim3(256), ldsBytes, ctx->stream, p, x->data, d_w, reinterpret_cast<const float4*>(d_epi),
                           out->data);
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

} // namespace

int make_conv2d_generic_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    if (g.preMode) return SNNHIP_E_UNSUPPORTED; // the fused-Pad address path exists in the MFMA kernel only
    if (g.normShift) return SNNHIP_E_UNSUPPORTED; // graph rule I: not in this kernel

    auto* plan = new GenericConvPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * g.kh * g.kw);
    plan->epi4 = epi4;
    GenericParams& p = plan->p;
    p.N = g.N; p.H = g.H; p.W = g.W; p.IC = g.IC; p.OC = g.OC; p.kh = g.kh; p.kw = g.kw; p.sh = g.sh; p.sw = g.sw;
    p.padx = g.padx; p.pady = g.pady; p.padMode = g.padMode; p.act = g.act; p.useBN = g.useBN; p.OH = g.OH; p.OW = g.OW;
    p.leaky = g.leaky;
    p.tileH = (TILE_H - 1) * g.sh + g.kh;
    p.tileW = (TILE_W - 1) * g.sw + g.kw;
    p.tileWp = p.tileW | 1; // odd pitch: rows/channels land on different banks
    p.tilesX = up_div(g.OW, TILE_W);
    p.ocBlocks = up_div(g.OC, OCB);
    const int taps = g.kh * g.kw;
    int icb = g.IC < 16 ? g.IC : 16;
    auto bytesFor = [&](int c) { return (static_cast<size_t>(p.tileH) * c * p.tileWp + static_cast<size_t>(taps) * c * OCB) * sizeof(float); };
    while (icb > 1 && bytesFor(icb) > 48 * 1024) icb = (icb + 1) / 2;
    p.ICB = icb;
    plan->ldsBytes = bytesFor(icb);
    if (plan->ldsBytes > 150 * 1024) {
        set_error("conv2d_generic: kernel %dx%d stride %d needs %zu B of LDS", g.kh, g.kw, g.sh, plan->ldsBytes);
        delete plan;
        return SNNHIP_E_UNSUPPORTED;
    }
    if (plan->ldsBytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(plan->ldsBytes));
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(%zu) failed: %s", plan->ldsBytes, hipGetErrorString(e));
            delete plan;
            return SNNHIP_E_HIP;
        }
    }
    // pack weights: [ocb][tap][IC][16]
    std::vector<float> wpk(static_cast<size_t>(p.ocBlocks) * taps * g.IC * OCB, 0.0f);
    for (int o = 0; o < g.OC; ++o)
        for (int i = 0; i < g.IC; ++i)
            for (int t = 0; t < taps; ++t)
                wpk[((static_cast<size_t>(o / OCB) * taps + t) * g.IC + i) * OCB + (o % OCB)] = w_oihw[(static_cast<size_t>(o) * g.IC + i) * taps + t];
    int rc = plan->upload(wpk.data(), wpk.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epi4.data(), epi4.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = g.N; plan->inDims[1] = g.H; plan->inDims[2] = g.W; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->flops = 2.0 * taps * g.IC * g.OC * static_cast<double>(g.OH) * g.OW * g.N;
    plan->bytes = 4.0 * (static_cast<double>(g.N) * g.H * g.W * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC +
                         static_cast<double>(g.OC) * g.IC * taps);
    char buf[256];
    snprintf(buf, sizeof(buf), "conv2d_generic_f32 k=%dx%d s=%d ic=%d oc=%d tile=32x8x16 icb=%d lds=%zuB", g.kh, g.kw, g.sh, g.IC, g.OC, icb,
             plan->ldsBytes);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
