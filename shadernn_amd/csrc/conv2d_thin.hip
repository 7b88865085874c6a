// conv2d_thin.hip -- Conv2D with at most 4 output channels (image-producing layers: Candy's 9x9 32->3 output conv, the
// unfused ESPCN 3x3 16->4) on v_mfma_f32_4x4x1_16B_f32.
//
// Replaces shadertemplate_vk_conv2d.comp:148-347 for those shapes.  A 16/32-wide MFMA tile would be >= 75 % padding with OC <= 4
// and the VALU kernel (conv2d_generic.hip) runs them at ~10 TFLOP/s.  The 4x4x1 instruction executes 16 independent 4x4 outer
// products per issue: block b = 4 adjacent output pixels, rows = the (up to) 4 output channels, one input channel per instruction:
//     D_b[oc][px] += W[oc][tap, ic] * x[px + tap][ic]
//   lane l  -> pixel l of the wave's 64-pixel batch (B operand: its own activation), output channel l%4 (A operand: its weight)
//   LDS     -> halo tile of the block, 16 channels per chunk, 64 B per pixel with the 16-byte-slot XOR of the fused ESPCN kernels
//              (conflict-free ds_read_b128 for 32 consecutive pixels); weights of the chunk as [tap][quad][oc][4 ic] so one
//              broadcast ds_read_b128 feeds 4 MFMAs
//   wave    -> two 64-pixel batches (2 rows of 32 each) share every weight read: 3 LDS reads per 8 MFMAs (the LDS would otherwise
//              be the bound: 2 reads per 4 MFMAs x 4 waves = 100 % of its issue rate)
//   block   -> 32 x 16 output pixels (4 waves) or 32 x 32 (8 waves, when the halo tile of a big kernel would leave one block per CU);
//              stride 1 only (both users), any kernel size / padding mode / activation
#include "epilogue.h"
#include "snnhip_internal.h"

#include <cstdlib>
#include <type_traits>

namespace snnhip {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TW = 32, ICC = 16; // output tile width (height = 4 rows per wave), channels per LDS chunk

struct ThinParams {
    int N, H, W, IC, OC, kh, kw, padx, pady, padMode, useBN, OH, OW;
    int tileH, tileW; // staged halo tile: TH + kh - 1, TW + kw - 1
    int tilesX, tilesY, nChunks;
    int TH;           // output tile height = 4 * waves per block (4 waves: 32x16, 8 waves: 32x32)
    int xFloats;      // floats of the activation tile in LDS (the weight slab follows)
};

// F16: half tensors and weights; a 16-byte slot holds 8 channels and feeds two v_mfma_f32_4x4x4f16 (4 channels each) instead of four
// v_mfma_f32_4x4x1f32, so an LDS chunk is 32 channels in the same 64 bytes per pixel.
template <bool SIMPLE, bool F16>
__global__ __launch_bounds__(512) void conv2d_thin_kernel(ThinParams p, ActCfg ac, const void* __restrict__ xv, const void* __restrict__ wp,
                                                         const float4* __restrict__ epi, void* __restrict__ yv) {
    typedef typename std::conditional<F16, _Float16, float>::type T;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    constexpr int CH = F16 ? 8 : 4;        // channels per 16-byte slot
    constexpr int CHUNK = 4 * CH;          // channels per LDS chunk
    const T* __restrict__ x = static_cast<const T*>(xv);
    T* __restrict__ y = static_cast<T*>(yv);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_x = smem;
    float* s_w = smem + p.xFloats;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int b = blockIdx.x;
    const int tx = b % p.tilesX;
    b /= p.tilesX;
    const int ty = b % p.tilesY;
    const int n = b / p.tilesY;
    const int ox0 = tx * TW, oy0 = ty * p.TH;
    const int nthreads = blockDim.x;
    const int ix0 = ox0 - p.padx, iy0 = oy0 - p.pady;
    const T* xn = x + static_cast<size_t>(n) * p.H * p.W * p.IC;
    const int taps = p.kh * p.kw;
    const bool vec4 = (p.IC % CH) == 0;

    // lane -> pixels: batch bt covers tile rows 4*wv + 2*bt + lane/32, column lane%32
    const int col = lane & 31, rsub = lane >> 5;
    int pix[2];
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) pix[bt] = (4 * wv + 2 * bt + rsub) * p.tileW + col;
    const float* wLane = s_w + (lane & 3) * 4;

    f32x4 acc[2][2]; // [batch][channel parity]: two chains per batch so consecutive MFMAs never wait on their own accumulator
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int k = 0; k < 2; ++k) acc[bt][k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int npix = p.tileH * p.tileW;
    for (int chunk = 0; chunk < p.nChunks; ++chunk) {
        const int ic0 = chunk * CHUNK;
        __syncthreads();
        // ---- stage the halo tile (padding resolved here) and this chunk's weights
        for (int e = tid; e < npix * 4; e += nthreads) {
            const int q = e & 3, pl = e >> 2;
            const int r = pl / p.tileW, c = pl - r * p.tileW;
            const int sy = resolve_coord(iy0 + r, p.H, p.padMode), sx = resolve_coord(ix0 + c, p.W, p.padMode);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int icq = ic0 + q * CH;
            if (sy >= 0 && sx >= 0 && icq < p.IC) {
                const T* src = xn + (static_cast<size_t>(sy) * p.W + sx) * p.IC + icq;
                if (vec4) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    T tmp[CH];
#pragma unroll
                    for (int j = 0; j < CH; ++j) tmp[j] = icq + j < p.IC ? src[j] : static_cast<T>(0.0f);
                    v = *reinterpret_cast<const float4*>(tmp);
                }
            }
            *reinterpret_cast<float4*>(s_x + pl * 16 + ((q ^ ((pl >> 2) & 3)) << 2)) = v;
        }
        const float4* wsrc = reinterpret_cast<const float4*>(wp) + static_cast<size_t>(chunk) * taps * 16;
        for (int e = tid; e < taps * 16; e += nthreads) reinterpret_cast<float4*>(s_w)[e] = wsrc[e];
        __syncthreads();

        // software pipeline over the taps: the 12 operand loads of tap t+1 are in flight while the 32 MFMAs of tap t issue
        f32x4 wq[4], xq[2][4];
        auto fetch = [&](int tap, int d) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                wq[q] = *reinterpret_cast<const f32x4*>(wLane + (tap * 4 + q) * 16);
#pragma unroll
                for (int bt = 0; bt < 2; ++bt) {
                    const int pl = pix[bt] + d;
                    xq[bt][q] = *reinterpret_cast<const f32x4*>(s_x + pl * 16 + ((q ^ ((pl >> 2) & 3)) << 2));
                }
            }
        };
        int fx = 0, rowoff = 0;
        fetch(0, 0);
#pragma unroll 1
        for (int tap = 0; tap < taps; ++tap) {
            f32x4 wc[4], xc[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                wc[q] = wq[q];
                xc[0][q] = xq[0][q];
                xc[1][q] = xq[1][q];
            }
            if (++fx == p.kw) {
                fx = 0;
                rowoff += p.tileW;
            }
            if (tap + 1 < taps) fetch(tap + 1, rowoff + fx);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (F16) {
                    const h4* wh = reinterpret_cast<const h4*>(&wc[q]);
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int bt = 0; bt < 2; ++bt)
                            acc[bt][k] = __builtin_amdgcn_mfma_f32_4x4x4f16(wh[k], reinterpret_cast<const h4*>(&xc[bt][q])[k], acc[bt][k], 0, 0, 0);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int bt = 0; bt < 2; ++bt) // accumulator index is a compile-time constant (a run-time one costs a select per register)
                            acc[bt][k & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc[q][k], xc[bt][q][k], acc[bt][k & 1], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lane holds its pixel's (up to) 4 output channels
    float4 e[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) e[o] = epi[o]; // table padded to 16 channels
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
        const int oy = oy0 + 4 * wv + 2 * bt + rsub, ox = ox0 + col;
        const f32x4 s = acc[bt][0] + acc[bt][1];
        float o4[4];
        // SiLU quirk: the 4-pixel group shares pixel 0's gate; pixel 0 of the aligned group is lane (col & ~3)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float v = epi_affine(s[o], e[o], p.useBN);
            if (SIMPLE) {
                v = apply_act<true>(ac, v, 0.0f);
            } else if (ac.act == SNNHIP_ACT_SILU_QUIRK) {
                const float own = epi_act(SNNHIP_ACT_SILU, ac.leaky, v, 0.0f);
                const float first = __shfl(own, lane & ~3, 64);
                v = (col & 3) == 0 ? own : epi_act(SNNHIP_ACT_SILU_QUIRK, ac.leaky, v, first);
            } else {
                v = epi_act(ac.act, ac.leaky, v, 0.0f);
            }
            o4[o] = v;
        }
        if (oy < p.OH && ox < p.OW) {
            T* dst = y + ((static_cast<size_t>(n) * p.OH + oy) * p.OW + ox) * p.OC;
            if (p.OC == 4 && !F16) {
                *reinterpret_cast<float4*>(dst) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            } else {
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < p.OC) dst[o] = static_cast<T>(o4[o]);
            }
        }
    }
}

struct ThinConvPlan : ConvPlanBase {
    ThinParams p;
    ActCfg ac;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    size_t ldsBytes = 0;
    int waves = 4;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "conv2d: expects 1 input, got %d", nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.H && x->w == p.W && x->c == p.IC, "conv2d: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h,
                       x->w, x->c, p.N, p.H, p.W, p.IC);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.OC, "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d",
                       out->n, out->h, out->w, out->c, p.N, p.OH, p.OW, p.OC);
        const dim3 grid(static_cast<unsigned>(p.tilesX * p.tilesY * p.N));
        const void* xv = x->data;
        const void* wv = d_w;
        void* yv = out->data;
        const float4* e4 = reinterpret_cast<const float4*>(d_epi);
        const bool simple = act_is_simple(ac.act);
        if (dtype == SNNHIP_F16) {
            if (simple) SNNHIP_LAUNCH((conv2d_thin_kernel<true, true>), grid, dim3(64 * waves), ldsBytes, ctx->stream, p, ac, xv, wv, e4, yv);
            else SNNHIP_LAUNCH((conv2d_thin_kernel<false, true>), grid, dim3(64 * waves), ldsBytes, ctx->stream, p, ac, xv, wv, e4, yv);
        } else {
            if (simple) SNNHIP_LAUNCH((conv2d_thin_kernel<true, false>), grid, dim3(64 * waves), ldsBytes, ctx->stream, p, ac, xv, wv, e4, yv);
            else SNNHIP_LAUNCH((conv2d_thin_kernel<false, false>), grid, dim3(64 * waves), ldsBytes, ctx->stream, p, ac, xv, wv, e4, yv);
        }
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

} // namespace

int make_conv2d_thin_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    if (g.preMode) return SNNHIP_E_UNSUPPORTED; // the fused-Pad address path exists in the MFMA kernel only
    if (g.normShift) return SNNHIP_E_UNSUPPORTED; // graph rule I: not in this kernel

    const char* force = snnhip::option("SNNHIP_CONV");
    if (force && strcmp(force, "thin") != 0) return SNNHIP_E_UNSUPPORTED; // generic / mfma forced
    if (g.OC > 4 || g.sh != 1 || g.sw != 1) return SNNHIP_E_UNSUPPORTED;
    if (!force && g.IC < 8) return SNNHIP_E_UNSUPPORTED; // a handful of input channels: the VALU kernel is as good
    ThinParams p{};
    p.N = g.N; p.H = g.H; p.W = g.W; p.IC = g.IC; p.OC = g.OC; p.kh = g.kh; p.kw = g.kw; p.padx = g.padx; p.pady = g.pady;
    p.padMode = g.padMode; p.useBN = g.useBN; p.OH = g.OH; p.OW = g.OW;
    const int taps = g.kh * g.kw;
    // 4 waves (32x16 tile) when at least 2 blocks fit a CU's LDS, else 8 waves on a 32x32 tile: a single wave per SIMD can only issue one
    // 4x4x1 MFMA every 16 cycles, two reach the pipe's 8
    auto ldsFor = [&](int waves) { return (static_cast<size_t>(4 * waves + g.kh - 1) * (TW + g.kw - 1) * 16 + static_cast<size_t>(taps) * 64) * sizeof(float); };
    int waves = 4;
    if (ldsFor(4) > 78 * 1024 && ldsFor(8) <= 150 * 1024) waves = 8;
    if (ldsFor(waves) > 150 * 1024) return SNNHIP_E_UNSUPPORTED;
    p.TH = 4 * waves;
    p.tileH = p.TH + g.kh - 1;
    p.tileW = TW + g.kw - 1;
    p.tilesX = up_div(g.OW, TW);
    p.tilesY = up_div(g.OH, p.TH);
    const bool f16 = g.dtype == SNNHIP_F16;
    const int CH = f16 ? 8 : 4, CHUNK = 4 * CH;
    p.nChunks = up_div(g.IC, CHUNK);
    p.xFloats = p.tileH * p.tileW * 16;
    const size_t ldsBytes = ldsFor(waves);
    auto* plan = new ThinConvPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * taps);
    plan->epi4 = epi4;
    plan->p = p;
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->ldsBytes = ldsBytes;
    plan->waves = waves;
    if (ldsBytes > 64 * 1024) {
        for (const void* fn : {reinterpret_cast<const void*>(conv2d_thin_kernel<true, false>), reinterpret_cast<const void*>(conv2d_thin_kernel<false, false>),
                               reinterpret_cast<const void*>(conv2d_thin_kernel<true, true>), reinterpret_cast<const void*>(conv2d_thin_kernel<false, true>)}) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsBytes));
            if (e != hipSuccess) {
                set_error("hipFuncSetAttribute(%zu) failed: %s", ldsBytes, hipGetErrorString(e));
                delete plan;
                return SNNHIP_E_HIP;
            }
        }
    }
    // weights: Wp[chunk][tap][slot(4)][oc(4)][CH channels] (16 bytes per (slot, oc)), zero for oc >= OC and ic >= IC
    std::vector<float> wpk(static_cast<size_t>(p.nChunks) * taps * 64, 0.0f);
    _Float16* wph = reinterpret_cast<_Float16*>(wpk.data());
    for (int o = 0; o < g.OC; ++o)
        for (int i = 0; i < g.IC; ++i)
            for (int t = 0; t < taps; ++t) {
                const int chunk = i / CHUNK, q = (i % CHUNK) / CH, k = i % CH;
                const size_t slot = ((static_cast<size_t>(chunk) * taps + t) * 4 + q) * 4 + o;
                const float wv = w_oihw[(static_cast<size_t>(o) * g.IC + i) * taps + t];
                if (f16) wph[slot * 8 + k] = static_cast<_Float16>(wv);
                else wpk[slot * 4 + k] = wv;
            }
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
    char buf[200];
    snprintf(buf, sizeof(buf), "conv2d_thin_mfma_%s k=%dx%d s=1 ic=%d oc=%d tile=32x%d chunk=%d lds=%zuB", f16 ? "f16_4x4x4" : "f32_4x4x1", g.kh, g.kw, g.IC,
             g.OC, p.TH, CHUNK, ldsBytes);
    plan->dtype = g.dtype;
    if (f16) plan->bytes *= 0.5;
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
