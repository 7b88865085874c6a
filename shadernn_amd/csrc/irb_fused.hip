// irb_fused.hip -- chain rule G: Conv2D 1x1 (expand) -> DepthwiseConv2D 3x3 (stride 1 | 2) -> Conv2D 1x1 (project) [-> Add with the block input]
// as ONE kernel: MobileNetV2's inverted-residual block (BASELINE configs[3]).  The expanded tensor (6x the block's channels) is what the
// unfused graph spends its time on -- written by the expand layer, read and re-written by the depthwise layer, read by the project layer:
// 0.24 GB of a 56x56x24 block's 0.25 GB at batch 32 -- and here it only ever exists as a 16-channel slice of one spatial tile in LDS.
//
// Same operator contracts as the separate layers (shadertemplate_vk_conv2d_1x1.comp:68-210, shadertemplate_vk_depthwise.comp:64-137:
// clipped taps = zero padding of the EXPANDED tensor, bias -> BN -> activation after each stage, vk_add.comp:41); BN is folded to
// (scale, shift) per channel on the host.
//
// One block = 256 threads = 4 waves, one output tile of TH x TW pixels (8x16 for stride 1, 8x8 for stride 2 and for the 7x7 / 14x14 maps):
//   x tile      the (TH-1)s+3 x (TW-1)s+3 halo tile of the block input, all C channels, staged ONCE in LDS as channel-quad planes
//               [quad][pixel] float4 (consecutive pixels = consecutive 16-byte slots, planes 256-byte aligned: conflict-free ds_read_b128 for
//               the 16 pixels x 4 quads of an MFMA operand); it also supplies the residual at the end
//   loop over 16-channel slices c of the expanded tensor, software-pipelined so that ONE barrier per slice suffices:
//     E(c+1)  expand, v_mfma_f32_16x16x4_f32: D[hc][px] = We[hc][ic] x[ic][px] over the halo tile's pixels (16 per MFMA tile, tiles dealt to
//             the waves), K permuted so that one float4 per operand feeds four MFMAs; act1(scale*D + shift), ZEROED outside the image (the
//             depthwise layer pads the expanded tensor with zeros, not with act1(shift)), written as quad planes to the other hidden buffer
//     D(c)    depthwise: lane (pixel n of a 16-pixel group, quad k) reads its 9 taps (ds_read_b128), 36 FMAs, act2 -> a float4 that IS the
//     P(c)    project MFMA's B operand (K = the slice's 16 channels, lane k owns 4k..4k+3): acc[co block][group] += Wp[co][hc] dw[hc][px]
//   weights of slice c+2 (expand) / c+1 (depthwise + project) arrive by LDS-DMA while slice c is computed (two buffers each; pre-packed on the
//   host as the LDS image, epilogue constants and depthwise taps riding in the same blobs)
//   epilogue: act3(scale*acc + shift) [+ x from the tile -> act4], 16-byte channel-contiguous stores.
#include <cstring>
#include <vector>

#include "epilogue.h"
#include "snnhip_internal.h"

namespace snnhip {

namespace {

// (scale, shift) per channel so that epilogue = act(acc * scale + shift): scale = bnScale, shift = bnScale * (bias - mean) + beta
std::vector<float> fold_epilogue(const std::vector<float>& epi4, int OC, int useBN) {
    std::vector<float> out(static_cast<size_t>(OC) * 2);
    for (int o = 0; o < OC; ++o) {
        const float bias = epi4[o * 4 + 0], sc = epi4[o * 4 + 1], mean = epi4[o * 4 + 2], beta = epi4[o * 4 + 3];
        out[o * 2 + 0] = useBN ? sc : 1.0f;
        out[o * 2 + 1] = useBN ? sc * (bias - mean) + beta : bias;
    }
    return out;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

struct IrbParams {
    int N, H, W, C, Ch, Co, OH, OW, s, padx, pady;
    int Cj;            // 16-channel groups of the (padded) input channels
    int NCB;           // 16-channel output blocks
    int TWs;           // log2(TW): 3 or 4; TH = 8
    int HH, HWd, HP;   // hidden halo tile: rows, columns, pixels
    int MT;            // ceil(HP / 16) expand MFMA tiles
    int tilesX, tilesY;
    int nChunks;       // ceil(Ch / 16)
    int wePieces, wpPieces;  // 256-float (1 KiB) pieces per slice blob
    int xPlane, hPlane;      // floats between quad planes of the x tile / a hidden buffer (multiples of 64)
    int offH, offWe, offWp, offMask; // LDS map in floats: x planes at 0
    int hasRes;
    ActCfg ac1, ac2, ac3, ac4;
};

constexpr int kMaxNCB = 20; // Co <= 320
constexpr int kMaxCj = 10;  // C <= 160

template <int G /* 16-pixel output groups per wave */, int NCBT /* compile-time bound on the output blocks */, int NW = 4 /* waves per block */>
__global__ __launch_bounds__(64 * NW) void irb_fused_kernel(IrbParams p, const float* __restrict__ x, const float4* __restrict__ weg, const float4* __restrict__ wpg,
                                                        const float4* __restrict__ epi3, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n16 = lane & 15, k = lane >> 4;
    const int TW = 1 << p.TWs;
    const int mt = blockIdx.x;
    const int tx = mt % p.tilesX, ty = (mt / p.tilesX) % p.tilesY, img = mt / (p.tilesX * p.tilesY);
    const int ox0 = tx * TW, oy0 = ty * 8;
    const int hx0 = ox0 * p.s - p.padx, hy0 = oy0 * p.s - p.pady; // image coordinates of the halo tile's origin

    float* const xs = smem;
    float* const hs = smem + p.offH;    // two hidden buffers of 4 planes
    float* const wes = smem + p.offWe;  // two expand blobs
    float* const wps = smem + p.offWp;  // two depthwise + project blobs
    float* const msk = smem + p.offMask;

    // LDS-DMA of slice blobs: wave w copies the 1 KiB pieces w, w + 4, ...
    auto dma = [&](const float4* g, float* dst, int pieces) {
        for (int pc = wave; pc < pieces; pc += NW) __builtin_amdgcn_global_load_lds(g + pc * 64 + lane, (lds_ptr)(dst + pc * 256), 16, 0, 0);
    };
    const size_t weStride = static_cast<size_t>(p.wePieces) * 64, wpStride = static_cast<size_t>(p.wpPieces) * 64; // float4 per slice
    // two buffers per blob kind, requested one interval ahead of their use.  (Rings of three with a distance of two and counted s_waitcnt were
    // measured: no faster on the blocks that keep two workgroups per CU, and the extra LDS cost b01 its second workgroup: 162 -> 265 us.)
    dma(weg, wes, p.wePieces);
    if (p.nChunks > 1) dma(weg + weStride, wes + p.wePieces * 256, p.wePieces);
    dma(wpg, wps, p.wpPieces);

    // ---- x tile: HP pixels x 4*Cj quads (channels past C and pixels outside the image are zero) + the inside-the-image mask
    {
        const int quads = 4 * p.Cj, cq = p.C >> 2;
        const int total = p.MT * 16 * quads;
        // batches of 8 elements per thread: all eight global loads are requested before the first LDS store (one element per loop iteration
        // serialised the HBM latency: 6-10 round trips per block with only two blocks per CU to hide them)
        for (int base = tid; base < total; base += 8 * 64 * NW) {
            float4 v[8];
            int lo[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int e = base + r * 64 * NW;
                v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                lo[r] = -1;
                if (e < total) {
                    const int hp = e / quads, q = e - hp * quads;
                    const int hy = hp / p.HWd, hx = hp - hy * p.HWd;
                    const int iy = hy0 + hy, ix = hx0 + hx;
                    const bool in = hp < p.HP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    if (in && q < cq) v[r] = *reinterpret_cast<const float4*>(x + ((static_cast<size_t>(img) * p.H + iy) * p.W + ix) * p.C + 4 * q);
                    lo[r] = q * p.xPlane + hp * 4;
                    if (q == 0) msk[hp] = in ? 1.0f : 0.0f;
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (lo[r] >= 0) *reinterpret_cast<float4*>(xs + lo[r]) = v[r];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the slice blobs requested above have landed in LDS ...
    __syncthreads();                                  // ... for every wave

    // ---- expand of slice c into hidden buffer c & 1 (wave-strided over the halo tile's 16-pixel MFMA tiles)
    auto expand = [&](int c) {
        const float* web = wes + (c & 1) * (p.wePieces * 256);
        float* hb = hs + (c & 1) * (4 * p.hPlane);
        float4 a[kMaxCj];
#pragma unroll
        for (int j = 0; j < kMaxCj; ++j)
            if (j < p.Cj) a[j] = *reinterpret_cast<const float4*>(web + j * 256 + lane * 4);
        const float4 sc = *reinterpret_cast<const float4*>(web + p.Cj * 256 + 4 * k);        // act1(scale * D + shift), channels 4k .. 4k+3 of the slice
        const float4 sh = *reinterpret_cast<const float4*>(web + p.Cj * 256 + 16 + 4 * k);
        // two MFMA tiles at a time: the second tile's MFMAs fill the 40-cycle dependent-accumulator latency of the first one's chain
        for (int t = wave; t < p.MT; t += 2 * NW) {
            const int px0 = t * 16 + n16;
            const bool two = t + NW < p.MT;         // wave-uniform
            const int px1 = two ? px0 + 16 * NW : px0;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < kMaxCj; ++j)
                if (j < p.Cj) {
                    const float4 b0 = *reinterpret_cast<const float4*>(xs + (4 * j + k) * p.xPlane + px0 * 4);
                    const float4 b1 = *reinterpret_cast<const float4*>(xs + (4 * j + k) * p.xPlane + px1 * 4);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b0.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b0.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b0.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b0.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b1.w, acc1, 0, 0, 0);
                }
            const float m0 = msk[px0], m1 = msk[px1];
            float4 h;
            h.x = apply_act<true>(p.ac1, fmaf(sc.x, acc0[0], sh.x), 0.f) * m0;
            h.y = apply_act<true>(p.ac1, fmaf(sc.y, acc0[1], sh.y), 0.f) * m0;
            h.z = apply_act<true>(p.ac1, fmaf(sc.z, acc0[2], sh.z), 0.f) * m0;
            h.w = apply_act<true>(p.ac1, fmaf(sc.w, acc0[3], sh.w), 0.f) * m0;
            *reinterpret_cast<float4*>(hb + k * p.hPlane + px0 * 4) = h;
            if (two) {
                h.x = apply_act<true>(p.ac1, fmaf(sc.x, acc1[0], sh.x), 0.f) * m1;
                h.y = apply_act<true>(p.ac1, fmaf(sc.y, acc1[1], sh.y), 0.f) * m1;
                h.z = apply_act<true>(p.ac1, fmaf(sc.z, acc1[2], sh.z), 0.f) * m1;
                h.w = apply_act<true>(p.ac1, fmaf(sc.w, acc1[3], sh.w), 0.f) * m1;
                *reinterpret_cast<float4*>(hb + k * p.hPlane + px1 * 4) = h;
            }
        }
    };

    // ---- this lane's output pixels: group g of the wave -> tile-local (row, column) -> top-left pixel of its 3x3 window in the halo tile
    int oyl[G], oxl[G], hp0[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int grp = wave * G + g;                 // 16 consecutive pixels of the tile in row-major order
        const int pl = grp * 16 + n16;
        oyl[g] = pl >> p.TWs;
        oxl[g] = pl & (TW - 1);
        hp0[g] = oyl[g] * p.s * p.HWd + oxl[g] * p.s;
    }
    f32x4 acc[NCBT][G];
#pragma unroll
    for (int cb = 0; cb < NCBT; ++cb)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[cb][g] = f32x4{0.f, 0.f, 0.f, 0.f};

    expand(0);
    __syncthreads();

    for (int c = 0; c < p.nChunks; ++c) {
        // slice blobs that the next interval needs: expand weights of c + 2 (their buffer was last read by E(c), before the barrier above),
        // depthwise + project weights of c + 1 (buffer last read by D/P(c - 1))
        if (c + 2 < p.nChunks) dma(weg + weStride * (c + 2), wes + (c & 1) * (p.wePieces * 256), p.wePieces);
        if (c + 1 < p.nChunks) dma(wpg + wpStride * (c + 1), wps + ((c + 1) & 1) * (p.wpPieces * 256), p.wpPieces);

        const float* hb = hs + (c & 1) * (4 * p.hPlane) + k * p.hPlane;
        const float* wpb = wps + (c & 1) * (p.wpPieces * 256);
        const float* dwb = wpb + p.NCB * 256; // [9 taps][16 channels], then scale[16], shift[16]
        float4 wd[9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) wd[tp] = *reinterpret_cast<const float4*>(dwb + tp * 16 + 4 * k);
        const float4 sc = *reinterpret_cast<const float4*>(dwb + 144 + 4 * k), sh = *reinterpret_cast<const float4*>(dwb + 160 + 4 * k);
        float4 dv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int fy = 0; fy < 3; ++fy)
#pragma unroll
                for (int fx = 0; fx < 3; ++fx) {
                    const float4 h = *reinterpret_cast<const float4*>(hb + (hp0[g] + fy * p.HWd + fx) * 4);
                    const float4 w = wd[fy * 3 + fx];
                    s4.x = fmaf(h.x, w.x, s4.x);
                    s4.y = fmaf(h.y, w.y, s4.y);
                    s4.z = fmaf(h.z, w.z, s4.z);
                    s4.w = fmaf(h.w, w.w, s4.w);
                }
            dv[g].x = apply_act<true>(p.ac2, fmaf(sc.x, s4.x, sh.x), 0.f);
            dv[g].y = apply_act<true>(p.ac2, fmaf(sc.y, s4.y, sh.y), 0.f);
            dv[g].z = apply_act<true>(p.ac2, fmaf(sc.z, s4.z, sh.z), 0.f);
            dv[g].w = apply_act<true>(p.ac2, fmaf(sc.w, s4.w, sh.w), 0.f);
        }
#pragma unroll
        for (int cb = 0; cb < NCBT; ++cb)
            if (cb < p.NCB) {
                const float4 a = *reinterpret_cast<const float4*>(wpb + cb * 256 + lane * 4);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, dv[g].x, acc[cb][g], 0, 0, 0);
                    acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, dv[g].y, acc[cb][g], 0, 0, 0);
                    acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, dv[g].z, acc[cb][g], 0, 0, 0);
                    acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, dv[g].w, acc[cb][g], 0, 0, 0);
                }
            }
        if (c + 1 < p.nChunks) expand(c + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds output channels 16 cb + 4k .. + 3 of its pixels
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int oy = oy0 + oyl[g], ox = ox0 + oxl[g];
        if (oy >= p.OH || ox >= p.OW) continue;
        float* yp = y + ((static_cast<size_t>(img) * p.OH + oy) * p.OW + ox) * p.Co;
        const int hpc = (oyl[g] + p.pady) * p.HWd + oxl[g] + p.padx; // residual (stride 1): the block input at the output pixel
#pragma unroll
        for (int cb = 0; cb < NCBT; ++cb) {
            const int co = cb * 16 + 4 * k;
            if (cb < p.NCB && co < p.Co) {
                const float4 sc = epi3[2 * (cb * 4 + k)], sh = epi3[2 * (cb * 4 + k) + 1];
                float4 o;
                o.x = apply_act<true>(p.ac3, fmaf(sc.x, acc[cb][g][0], sh.x), 0.f);
                o.y = apply_act<true>(p.ac3, fmaf(sc.y, acc[cb][g][1], sh.y), 0.f);
                o.z = apply_act<true>(p.ac3, fmaf(sc.z, acc[cb][g][2], sh.z), 0.f);
                o.w = apply_act<true>(p.ac3, fmaf(sc.w, acc[cb][g][3], sh.w), 0.f);
                if (p.hasRes) {
                    const float4 r = *reinterpret_cast<const float4*>(xs + (co >> 2) * p.xPlane + hpc * 4);
                    o.x = apply_act<true>(p.ac4, o.x + r.x, 0.f);
                    o.y = apply_act<true>(p.ac4, o.y + r.y, 0.f);
                    o.z = apply_act<true>(p.ac4, o.z + r.z, 0.f);
                    o.w = apply_act<true>(p.ac4, o.w + r.w, 0.f);
                }
                *reinterpret_cast<float4*>(yp + co) = o;
            }
        }
    }
}

struct IrbPlan : snnhip_plan {
    IrbParams p;
    float* d_we = nullptr;
    float* d_wp = nullptr;
    float* d_e3 = nullptr;
    size_t ldsBytes = 0;
    dim3 grid;
    int G = 1, ncbt = 2, threads = 256;
    void (*kernel)(IrbParams, const float*, const float4*, const float4*, const float4*, float*) = nullptr;

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "inverted-residual block: expects 1 input (the block input is also the residual), got %d", nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.H && x->w == p.W && x->c == p.C, "irb: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h, x->w, x->c,
                       p.N, p.H, p.W, p.C);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.Co, "irb: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n,
                       out->h, out->w, out->c, p.N, p.OH, p.OW, p.Co);
        hipLaunchKernelGGL(kernel, grid, dim3(static_cast<unsigned>(threads)), ldsBytes, ctx->stream, p, x->data, reinterpret_cast<const float4*>(d_we),
                           reinterpret_cast<const float4*>(d_wp), reinterpret_cast<const float4*>(d_e3), out->data);
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

typedef void (*IrbFn)(IrbParams, const float*, const float4*, const float4*, const float4*, float*);
template <int G>
IrbFn pick_irb(int ncb) {
    if (ncb <= 2) return irb_fused_kernel<G, 2>;
    if (ncb <= 4) return irb_fused_kernel<G, 4>;
    if (ncb <= 6) return irb_fused_kernel<G, 6>;
    if (G == 1 && ncb <= 10) return irb_fused_kernel<1, 10>;
    if (G == 1 && ncb <= 20) return irb_fused_kernel<1, 20>;
    return nullptr;
}
} // namespace

// expand / dw / project: the three per-layer plans (borrowed; only read here); add: the residual Add plan or nullptr.
int make_irb_plan(snnhip_ctx* ctx, snnhip_plan* expandPlan, snnhip_plan* dwPlan, snnhip_plan* projectPlan, snnhip_plan* addPlan, snnhip_plan** out) {
    if (snnhip::option("SNNHIP_NO_IRB_FUSION")) return SNNHIP_E_UNSUPPORTED;
    const char* irbMode = snnhip::option("SNNHIP_IRB_FUSION"); // "all": also the 14x14 / 7x7 blocks, where the separate layers are faster (tools/bench_irb.py)
    auto* ce = dynamic_cast<ConvPlanBase*>(expandPlan);
    auto* cd = dynamic_cast<ConvPlanBase*>(dwPlan);
    auto* cp = dynamic_cast<ConvPlanBase*>(projectPlan);
    auto* ad = addPlan ? dynamic_cast<EltwisePlanBase*>(addPlan) : nullptr;
    if (!ce || !cd || !cp || ce->depthwise || !cd->depthwise || cp->depthwise || (addPlan && (!ad || ad->mode != 0))) return SNNHIP_E_UNSUPPORTED;
    const ConvGeom &ge = ce->g, &gd = cd->g, &gp = cp->g;
    auto pointwise = [](const ConvGeom& g) { return g.kh == 1 && g.kw == 1 && g.sh == 1 && g.sw == 1 && g.preMode == 0 && g.addAct < 0 && g.dtype == SNNHIP_F32; };
    if (!pointwise(ge) || !pointwise(gp) || gd.dtype != SNNHIP_F32 || gd.kh != 3 || gd.kw != 3 || gd.sh != gd.sw || gd.sh < 1 || gd.sh > 2 || gd.preMode != 0)
        return SNNHIP_E_UNSUPPORTED;
    if (gd.padMode != SNNHIP_PAD_CONSTANT && gd.padMode != SNNHIP_PAD_NONE) return SNNHIP_E_UNSUPPORTED;
    if (ge.OC != gd.IC || gd.OC != gp.IC || ge.N != gd.N || gd.N != gp.N || ge.OH != gd.H || ge.OW != gd.W || gd.OH != gp.H || gd.OW != gp.W || ge.OH != ge.H ||
        ge.OW != ge.W || gp.OH != gp.H || gp.OW != gp.W)
        return SNNHIP_E_UNSUPPORTED;
    if (gd.padx < 0 || gd.padx > 2 || gd.pady < 0 || gd.pady > 2) return SNNHIP_E_UNSUPPORTED;
    const int C = ge.IC, Ch = ge.OC, Co = gp.OC, s = gd.sh;
    // Where it pays: the blocks whose expanded tensor is big (MobileNetV2 b01-b06, 112x112 .. 28x28 inputs).  On the 14x14 / 7x7 maps the
    // separate layers win (measured at batch 64: 49 vs 88 us for 64->384->64 @14x14, 72 vs 245 us for 160->960->160 @7x7): few tiles per
    // image, 24-60 slices of one barrier each, weights that no longer fit beside the x tile.
    const bool fuseAll = irbMode && strcmp(irbMode, "all") == 0;
    if (!fuseAll && ge.H * ge.W < 28 * 28) return SNNHIP_E_UNSUPPORTED;
    if (C % 4 || Ch % 4 || Co % 4 || C > 16 * kMaxCj || Co > 16 * kMaxNCB) return SNNHIP_E_UNSUPPORTED;
    const int acts[4] = {ge.act, gd.act, gp.act, ad ? ad->d.act : 0};
    for (int a : acts)
        if (!act_is_simple(a)) return SNNHIP_E_UNSUPPORTED;
    if (ad && (s != 1 || C != Co || ad->d.N != gp.N || ad->d.H != gp.OH || ad->d.W != gp.OW || ad->d.C != Co)) return SNNHIP_E_UNSUPPORTED;
    if (static_cast<double>(ge.N) * ge.H * ge.W * std::max(C, Ch) >= 2147483647.0) return SNNHIP_E_UNSUPPORTED;

    IrbParams p = {};
    p.N = ge.N; p.H = ge.H; p.W = ge.W; p.C = C; p.Ch = Ch; p.Co = Co; p.OH = gd.OH; p.OW = gd.OW; p.s = s; p.padx = gd.padx; p.pady = gd.pady;
    p.Cj = up_div(C, 16);
    p.NCB = up_div(Co, 16);
    // 8x16 output tiles (two pixel groups per wave) for stride-1 blocks on maps that have the columns, 8x8 otherwise
    const bool wide = s == 1 && p.OW > 8 && p.NCB <= 6;
    p.TWs = wide ? 4 : 3;
    const int TW = 1 << p.TWs, TH = 8;
    p.HH = (TH - 1) * s + 3;
    p.HWd = (TW - 1) * s + 3;
    p.HP = p.HH * p.HWd;
    p.MT = up_div(p.HP, 16);
    p.tilesX = up_div(p.OW, TW);
    p.tilesY = up_div(p.OH, TH);
    p.nChunks = up_div(Ch, 16);
    p.wePieces = p.Cj + 1;
    p.wpPieces = p.NCB + 1;
    p.xPlane = round_up(p.MT * 16 * 4, 64);
    p.hPlane = p.xPlane;
    p.offH = 4 * p.Cj * p.xPlane;
    p.offWe = p.offH + 2 * 4 * p.hPlane;
    p.offWp = p.offWe + 2 * p.wePieces * 256;
    p.offMask = p.offWp + 2 * p.wpPieces * 256;
    p.hasRes = ad ? 1 : 0;
    p.ac1 = make_act_cfg(ge.act, ge.leaky);
    p.ac2 = make_act_cfg(gd.act, gd.leaky);
    p.ac3 = make_act_cfg(gp.act, gp.leaky);
    p.ac4 = make_act_cfg(ad ? ad->d.act : 0, ad ? ad->d.leaky : 0.0f);
    const size_t lds = static_cast<size_t>(p.offMask + p.MT * 16) * sizeof(float);
    if (lds > 160 * 1024) return SNNHIP_E_UNSUPPORTED;
    // ... and it needs two workgroups per CU to hide its one-barrier-per-slice structure: 56x56 24->144->32 stride 2 (91 KB) measured 138 us
    // fused against 89 us for the separate layers at batch 64
    if (!fuseAll && lds > 80 * 1024) return SNNHIP_E_UNSUPPORTED;
    // (512-thread blocks with one pixel group per wave were measured on the 8x16 tiles: slower, 494 -> 596 us on b02 at batch 256 -- the kernel is
    // bound by its VALU work per slice (PMC: 2540 VALU and 162 MFMA instructions per wave on b01), not by latency that more waves could hide)
    IrbFn fn = wide ? pick_irb<2>(p.NCB) : pick_irb<1>(p.NCB);
    if (!fn) return SNNHIP_E_UNSUPPORTED;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
        set_error("irb_fused: hipFuncSetAttribute(%zu) failed", lds);
        return SNNHIP_E_HIP;
    }

    // ---- slice blobs (the kernel's LDS images)
    const std::vector<float> e1 = fold_epilogue(ce->epi4, Ch, ge.useBN), e2 = fold_epilogue(cd->epi4, Ch, gd.useBN), e3 = fold_epilogue(cp->epi4, Co, gp.useBN);
    std::vector<float> we(static_cast<size_t>(p.nChunks) * p.wePieces * 256, 0.0f), wp(static_cast<size_t>(p.nChunks) * p.wpPieces * 256, 0.0f);
    for (int c = 0; c < p.nChunks; ++c) {
        float* wb = we.data() + static_cast<size_t>(c) * p.wePieces * 256;
        float* pb = wp.data() + static_cast<size_t>(c) * p.wpPieces * 256;
        for (int m = 0; m < 16; ++m) {
            const int hc = 16 * c + m;
            if (hc >= Ch) continue;
            // expand: [j][lane = 16 kk + m] float4 {We[hc][16 j + 4 kk + jj]}
            for (int ic = 0; ic < C; ++ic) {
                const int j = ic / 16, kk = (ic % 16) / 4, jj = ic % 4;
                wb[j * 256 + (kk * 16 + m) * 4 + jj] = ce->w_oihw[static_cast<size_t>(hc) * C + ic];
            }
            wb[p.Cj * 256 + m] = e1[2 * hc];
            wb[p.Cj * 256 + 16 + m] = e1[2 * hc + 1];
            // depthwise taps [tap][16], scale[16], shift[16]
            for (int tp = 0; tp < 9; ++tp) pb[p.NCB * 256 + tp * 16 + m] = cd->w_oihw[static_cast<size_t>(hc) * 9 + tp];
            pb[p.NCB * 256 + 144 + m] = e2[2 * hc];
            pb[p.NCB * 256 + 160 + m] = e2[2 * hc + 1];
        }
        // project: [cb][lane = 16 kk + m] float4 {Wp[co = 16 cb + m][hc = 16 c + 4 kk + jj]}
        for (int co = 0; co < Co; ++co)
            for (int q = 0; q < 16; ++q) {
                const int hc = 16 * c + q;
                if (hc >= Ch) continue;
                pb[(co / 16) * 256 + ((q / 4) * 16 + co % 16) * 4 + q % 4] = cp->w_oihw[static_cast<size_t>(co) * Ch + hc];
            }
    }
    // final epilogue: per (cb, k) {scale float4, shift float4}
    std::vector<float> e3p(static_cast<size_t>(p.NCB) * 4 * 8, 0.0f);
    for (int co = 0; co < Co; ++co) {
        e3p[(co / 4) * 8 + co % 4] = e3[2 * co];
        e3p[(co / 4) * 8 + 4 + co % 4] = e3[2 * co + 1];
    }

    auto* plan = new IrbPlan();
    plan->ctx = ctx;
    plan->p = p;
    plan->kernel = fn;
    plan->threads = 256;
    plan->ldsBytes = lds;
    plan->grid = dim3(p.tilesX * p.tilesY * p.N);
    plan->dtype = SNNHIP_F32;
    int rc = plan->upload(we.data(), we.size(), &plan->d_we);
    if (rc == SNNHIP_OK) rc = plan->upload(wp.data(), wp.size(), &plan->d_wp);
    if (rc == SNNHIP_OK) rc = plan->upload(e3p.data(), e3p.size(), &plan->d_e3);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    memcpy(plan->inDims, expandPlan->inDims, sizeof(plan->inDims));
    memcpy(plan->outDims, projectPlan->outDims, sizeof(plan->outDims));
    plan->flops = ce->flops + cd->flops + cp->flops;
    plan->bytes = ce->bytes + cd->bytes + cp->bytes + (addPlan ? addPlan->bytes : 0.0); // unfused accounting of the layers it replaces
    const double fusedBytes = 4.0 * (static_cast<double>(p.N) * p.H * p.W * C + static_cast<double>(p.N) * p.OH * p.OW * Co + static_cast<double>(Ch) * (C + Co + 9));
    char buf[320];
    snprintf(buf, sizeof(buf), "irb_fused_mfma_f32_16x16x4 [conv1x1 %d->%d + depthwise3x3 s%d + conv1x1 %d->%d%s] tile=8x%dpx halo=%dx%d slices=%d threads=%d lds=%zuB hbm_bytes=%.6g kernel=irb_fused_kernel",
             C, Ch, s, Ch, Co, addPlan ? " + add" : "", TW, p.HH, p.HWd, p.nChunks, 256, lds, fusedBytes);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
