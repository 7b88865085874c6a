// irb_fused.hip -- chain rule G: Conv2D 1x1 (expand) -> DepthwiseConv2D 3x3 (stride 1 | 2) -> Conv2D 1x1 (project) [-> Add with the block input]
// as ONE kernel: MobileNetV2's inverted-residual block (BASELINE configs[3]).  The expanded tensor (6x the block's channels) is what the
// unfused graph spends its time on -- written by the expand layer, read and re-written by the depthwise layer, read by the project layer:
// 0.24 GB of a 56x56x24 block's 0.25 GB at batch 32 -- and here it only ever exists as a 16-channel slice of one spatial tile in LDS.
//
// Same operator contracts as the separate layers (shadertemplate_vk_conv2d_1x1.comp:68-210, shadertemplate_vk_depthwise.comp:64-137:
// clipped taps = zero padding of the EXPANDED tensor, bias -> BN -> activation after each stage, vk_add.comp:41); BN is folded to
// (scale, shift) per channel on the host.
//
// One WAVE = one output tile of 2G x 8 pixels (G = 2 for stride 1, 1 for stride 2), no barrier in the kernel (see irb_wave_kernel):
//   x tile      the wave's (2G-1)s+3 x 7s+3 halo tile of the block input, all C channels, staged once in the wave's LDS slice as channel-quad planes
//               [quad][pixel] float4 (consecutive pixels = consecutive 16-byte slots, planes 256-byte aligned: conflict-free ds_read_b128 for
//               the 16 pixels x 4 quads of an MFMA operand); it also supplies the residual at the end
//   loop over 16-channel slices c of the expanded tensor:
//     E(c)    expand, v_mfma_f32_16x16x4_f32: D[hc][px] = We[hc][ic] x[ic][px] over the halo tile's pixels (16 per MFMA tile, two tiles in
//             flight), K permuted so that one float4 per operand feeds four MFMAs; act1(scale*D + shift), ZEROED outside the image (the
//             depthwise layer pads the expanded tensor with zeros, not with act1(shift)), written as quad planes to the wave's hidden slice
//     D(c)    depthwise: lane (pixel n of a 16-pixel group, quad k) reads its 9 taps (ds_read_b128), 36 FMAs, act2 -> a float4 that IS the
//     P(c)    project MFMA's B operand (K = the slice's 16 channels, lane k owns 4k..4k+3): acc[co block][group] += Wp[co][hc] dw[hc][px]
//   weights: pre-packed on the host in operand order (one 1 KiB piece per MFMA operand set, epilogue constants and depthwise taps riding in the same
//   blobs), read by every wave straight from L1 / L2
//   epilogue: act3(scale*acc + shift) [+ x from the tile -> act4], 16-byte channel-contiguous stores.
#include <cstdio>
#include <cstring>
#include <vector>

#include "epilogue.h"
#include "snnhip_internal.h"

#ifndef SNNHIP_IRB_FOLD_BN
#define SNNHIP_IRB_FOLD_BN 1 // (round 6) the expand / depthwise layers' folded BN scale is multiplied into their weights at plan creation and the accumulators start from the shift
#endif                      // (the MFMA's C operand, the tap sum's first addend): two packed FMAs fewer per pixel tile and slice in E and in D.  0: the round-5 form (A/B builds)
#ifndef SNNHIP_IRBI_FOLD_BN
#define SNNHIP_IRBI_FOLD_BN 0 // ... but not in irb_image_kernel: same-box A/B with the fold b14 108.3 vs 104.4 us, c4's nine image launches 839 vs 798 us (one wave per SIMD at 350 - 500
#endif                       // registers: the allocation the compiler finds for the shorter epilogue is the slower one); its plans keep unscaled weights

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
    int offH, offWe, offWp, offMask; // LDS map in floats: x planes at 0 (wave kernel: per wave; offWe = its mask, offMask = floats per wave)
    int hasRes;
    int tail8;         // C % 16 == 8: the expand layer's last 8 channels are TWO MFMAs (k lane kk = channel 16 j + kk, then 16 j + 4 + kk) with 4-byte operand reads
                       // from quad planes 4 j and 4 j + 1, instead of four MFMAs half of whose k lanes multiply zeros (MobileNetV2 b02 / b03: 24 channels, 6 instead of 8)
    int noExpand;      // DepthwiseConv2D -> Conv2D 1x1 without an expand layer in front (MobileNetV2's first block): the 'hidden' slice is the x tile itself
    unsigned magicQuads, magicHWd; // ceil(2^32 / d) for d = 4 Cj and HWd: the staging's two divisions as mul-hi (operands < 2^16: exact)
    // stem mode (stemK > 0): the 'expand' layer is a 3x3 convolution of a 3-channel image (MobileNetV2's Conv2D 3x3 s2 3->32 in front of its first,
    // expansion-less block).  x is the image; the x tile holds, per halo pixel (= stem output pixel), the stemK = 27 image values under the 3x3
    // window as 'channels' tap*3 + c (im2col while staging), and the expand MFMAs are the stem convolution.  (H, W) stay the dims of the tensor the
    // depthwise layer reads = the stem's output.
    int stemK, stemS, stemPadX, stemPadY, IH, IW;
    int rawH, rawW3;     // the image patch under the halo tile: rows, floats per row (3 per pixel)
    unsigned magicRawW3; // ceil(2^32 / rawW3)
    ActCfg ac1, ac2, ac3, ac4;
};

constexpr int kMaxNCB = 20; // Co <= 320
constexpr int kMaxCj = 10;  // C <= 160

// The kernel is bound by VALU issue (PMC: 2540 VALU and 162 MFMA instructions per wave on MobileNetV2's b01), most of it the expand epilogue over
// the halo tile and the depthwise epilogue: act(scale * D + shift) [* mask].  The general simple-activation form is three instructions
// (mul, max, med3); ReLU6 -- every expand / depthwise activation of MobileNetV2 -- is ONE (med3(v, 0, 6)), so the hot epilogues are instantiated
// for it (R6), and the border mask is only applied by tiles that touch the image border.
template <bool R6>
__device__ __forceinline__ float irb_act(const ActCfg& a, float v) {
    return R6 ? __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f) : apply_act<true>(a, v, 0.f);
}
// ReLU6 of a hidden channel the split-precision form keeps scaled by a power of two: the bound is 6 x that power (same instruction, a register instead of the constant)
__device__ __forceinline__ float irb_relu_to(float v, float bound) { return __builtin_amdgcn_fmed3f(v, 0.0f, bound); }

// ---- split-precision form of the two pointwise stages (round 6, S16): every fp32 operand x is carried as two halves, x = hi + lo (hi = fp16(x), lo = fp16(x - hi),
// |x - hi - lo| <= 2^-22 |x|), and a product w x is the three f16 MFMA products wh xh + wh xl + wl xh accumulated in fp32 (the dropped wl xl is 2^-22 of the product):
// ONE v_mfma_f32_16x16x32_f16 whose K axis carries (channel, hi | lo) -- A = [wh | wh], B = [xh | xl]: the 16-byte slot a lane read as four fp32 channels is now the
// same four channels as [hi x 4 | lo x 4] -- plus ONE v_mfma_f32_16x16x16_f16 (A = wl, B = xh = the slot's lower half) replace FOUR v_mfma_f32_16x16x4_f32:
// 2 x ~18 cycles of the matrix pipe instead of 4 x 32 (tools/ubench_mfma_f16.hip), same accumulator layout.  Ranges: the block input is scaled per tile by a power of
// two when its largest magnitude leaves [2^-2, 2^15) (split_tile_scale; undone in the expand epilogue), the depthwise output is ReLU6's [0, 6]; weight rows are normalised by powers of
// two on the host (undone in the same scales); lo parts below fp16's normal range lose bits that are 2^-25 of the row's largest magnitude.  The depthwise stage, all
// epilogues, the reduction and the residual (re-read from global memory: the LDS tile holds the split form) are fp32 as before.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f16x8 split_f16x3(float d0, float d1, float d2, float d3) {
    const f16x2 h01 = __builtin_convertvector(f32x2v{d0, d1}, f16x2), h23 = __builtin_convertvector(f32x2v{d2, d3}, f16x2); // (round to nearest even; the residuals are exact)
    const float r0 = d0 - static_cast<float>(h01[0]), r1 = d1 - static_cast<float>(h01[1]), r2 = d2 - static_cast<float>(h23[0]), r3 = d3 - static_cast<float>(h23[1]);
    const f16x2 l01 = __builtin_convertvector(f32x2v{r0, r1}, f16x2), l23 = __builtin_convertvector(f32x2v{r2, r3}, f16x2);
    return f16x8{h01[0], h01[1], h23[0], h23[1], l01[0], l01[1], l23[0], l23[1]};
}

// The power of two a tile is multiplied by before it is split (mb = bit pattern of its largest magnitude): none while that magnitude is in [2^-2, 2^15) -- fp16 holds the
// hi parts, and the lo parts of the elements that matter stay normal numbers (an element's lo part is 2^-11 of it; what falls below fp16's normal range is resolved to
// 2^-25, i.e. 2^-23 of the tile's largest magnitude or better) --, otherwise the power that brings it into [2^14, 2^15).  Returns whether the tile is scaled.
__device__ __forceinline__ bool split_tile_scale(int mb, float& sx, float& sxInv) {
    const int ex = (mb >> 23) & 255;
    const bool scaled = mb != 0 && (ex >= 127 + 15 || ex < 127 - 2);
    const int sb = scaled ? min(268 - ex, 227) : 127;
    sx = __int_as_float(sb << 23);
    sxInv = __int_as_float((254 - sb) << 23);
    return scaled;
}

// Wave-autonomous: every WAVE owns one small output tile (4x8 pixels for stride 1, 2x8 for stride 2) from the x tile to the store, in its own
// slice of LDS, and the kernel has NO barrier at all.  Its predecessor gave a 256-thread block one 8x16 / 8x8 tile and synchronised the four
// waves once per 16-channel slice, with ~100 MFMAs of work between barriers: b01 / b02 of MobileNetV2 ran 528 / 495 us at batch 256 with the
// matrix pipe 17 % busy and the waves waiting half of their cycles (PMC); shrinking its VALU work (ReLU6 as one med3, border mask only on
// border tiles) did not move it -- it waited, it did not issue.  Here a wave's expand -> depthwise -> project chain depends only on its own
// LDS traffic (in order per wave), the other waves of the SIMD fill its latencies, and the weights of a slice come straight from the packed
// blobs in L1 / L2 (16 float4 per lane and slice).  What decides the speed is how many waves a CU's LDS holds: 8x8 tiles per wave (22 KB) ran
// no faster than the block kernel, 4x8 / 2x8 tiles (12 KB, 12 waves per CU) 1.45x faster despite 1.9x instead of 1.56x halo work in the expand.
// One lane = (pixel n16 of a 16-pixel group, channel quad k); weights pre-packed on the host in the MFMA operand order (K permuted so that one
// float4 per operand feeds four v_mfma_f32_16x16x4_f32).
template <int G /* 16-pixel output groups per wave: 4 = 8x8 tile, 2 = 4x8, 1 = 2x8 */, int NCBT /* compile-time bound on the output blocks */, int CJT /* ... on Cj */,
          bool R6 /* expand and depthwise activations are ReLU6 */, bool STEM = false /* IrbParams::stemK: the expand layer is a 3x3 convolution of an RGB image */,
          bool S16 = false /* split-precision pointwise stages (irb_image_kernel's note); the wave's x tile is split in place behind the staging */>
__global__ __launch_bounds__(256, 2) void irb_wave_kernel(IrbParams p, const float* __restrict__ x, const float4* __restrict__ weg, const float4* __restrict__ wpg,
                                                       const float4* __restrict__ epi3, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n16 = lane & 15, k = lane >> 4;
    const int NWv = blockDim.x >> 6;
#ifdef SNNHIP_IRB_TRACE // experiment builds (tools/exp_one.sh)
    const bool itrace = blockIdx.x == 1000 && lane == 0 && (wave == 0 || wave == 2);
    unsigned long long istamp[4] = {};
    if (itrace) istamp[0] = __builtin_readcyclecounter();
#endif
    constexpr int TW = 8, TH = G * 2; // 16 G pixels
    const int mt = blockIdx.x * NWv + wave;
    if (mt >= p.tilesX * p.tilesY * p.N) return; // (no barrier in this kernel)
    const int tx = mt % p.tilesX, ty = (mt / p.tilesX) % p.tilesY, img = mt / (p.tilesX * p.tilesY);
    const int ox0 = tx * TW, oy0 = ty * TH;
    const int hx0 = ox0 * p.s - p.padx, hy0 = oy0 * p.s - p.pady;
    const bool border = hx0 < 0 || hy0 < 0 || hx0 + p.HWd > p.W || hy0 + p.HH > p.H; // (wave-uniform)

    float* const xs = smem + wave * p.offMask; // offMask doubles as the per-wave LDS size in floats: [x planes | hidden planes | mask]
    float* const hs = xs + p.offH;
    float* const msk = xs + p.offWe;

    // ---- x tile of this wave: HP pixels x 4*Cj quads (channels past C and pixels outside the image are zero) + the inside-the-image mask
    if constexpr (STEM) {
        // stem mode, two steps.  (a) The raw image patch under the halo tile -- rawH rows of rawW3 = 3 x pixels consecutive floats (interleaved RGB), zero
        // outside the image = the stem's zero padding -- is copied into the (still unused) hidden-slice region: row segments are contiguous in memory,
        // a wave instruction covers a whole segment.  (b) The x tile is built from it by LDS reads: 'channel' ch = 3 tap + c of halo pixel (hy, hx) is
        // raw[(hy stemS + fy) rawW3 + 3 hx stemS + (ch - 9 fy)], fy = ch / 9: the nine values of one window row are adjacent in the patch.
        // (Gathering the 27 values per pixel straight from global memory -- 32 predicated scalar loads per lane -- took 14.8k of the wave's 27k cycles.)
        // The kernel is bound by instruction issue (irb_act's note), so both steps walk their index spaces incrementally -- a lane's elements are 64
        // apart: (row, col) and (hy, hx) advance by an add and a wrap -- instead of dividing per element, and (b) uses that the tile has 8 quads:
        // a lane's quad, hence its four tap offsets, never change.
        const int ry0 = hy0 * p.stemS - p.stemPadY, rx0 = (hx0 * p.stemS - p.stemPadX) * 3, rowF = p.IW * 3;
        const int rawTotal = p.rawH * p.rawW3;
        const float* const ximg = x + static_cast<size_t>(img) * p.IH * rowF;
        {
            int row = static_cast<int>(__umulhi(static_cast<unsigned>(lane), p.magicRawW3)), col = lane - row * p.rawW3; // (rawW3 >= 33: host)
            for (int base = lane; base < rawTotal; base += 16 * 64) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int iy = ry0 + row, fx = rx0 + col;
                    const bool in = base + r * 64 < rawTotal && iy >= 0 && iy < p.IH && fx >= 0 && fx < rowF;
                    const float t = ximg[in ? iy * rowF + fx : 0];
                    v[r] = in ? t : 0.0f;
                    col += 64;
                    if (col >= p.rawW3) { col -= p.rawW3; ++row; }
                    if (col >= p.rawW3) { col -= p.rawW3; ++row; }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (base + r * 64 < rawTotal) hs[base + r * 64] = v[r];
            }
        }
#ifdef SNNHIP_IRB_TRACE
        if (itrace) {
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            istamp[3] = __builtin_readcyclecounter();
        }
#endif
        {
            const int q = lane >> 3, rowStep = p.rawW3 - 9; // (eight consecutive lanes = eight consecutive pixels of one quad: conflict-free ds_write_b128, see below)
            int tapOff[4];
#pragma unroll
            for (int cm = 0; cm < 4; ++cm) {
                const int ch = min(4 * q + cm, 26);
                tapOff[cm] = ((ch * 57) >> 9) * rowStep + ch; // fy (rawW3 - 9) + ch, fy = ch / 9 for ch < 32
            }
            const bool k3 = 4 * q + 3 < p.stemK; // only 'channel' 27 (quad 6, component 3) and quads 7 are padding
            const bool kq = 4 * q < p.stemK;
            const int iters = p.MT * 2; // MT 16 pixels x 8 quads / 64 lanes
            int hp = lane & 7;
            int hy = 0, hx = hp; // (HWd >= 8: the first eight pixels are in row 0, one wrap per step)
            float* xq = xs + q * p.xPlane + hp * 4;
#pragma unroll 4 // (iters is even: MT 16 pixels; four iterations' LDS reads in flight)
            for (int it = 0; it < iters; ++it) {
                const bool real = hp < p.HP; // the padding pixels of the last MFMA tile: zeros
                const float* rp = hs + (real ? hy * p.stemS * p.rawW3 + hx * p.stemS * 3 : 0);
                const float t0 = rp[tapOff[0]], t1 = rp[tapOff[1]], t2 = rp[tapOff[2]], t3 = rp[tapOff[3]];
                const bool on = real && kq;
                *reinterpret_cast<float4*>(xq) = make_float4(on ? t0 : 0.0f, on ? t1 : 0.0f, on ? t2 : 0.0f, (on && k3) ? t3 : 0.0f);
                if (q == 0) {
                    const int sy = hy0 + hy, sx = hx0 + hx;
                    msk[hp] = (real && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) ? 1.0f : 0.0f;
                }
                hp += 8;
                xq += 32;
                hx += 8;
                if (hx >= p.HWd) { hx -= p.HWd; ++hy; }
            }
        }
    } else {
        const int quads = 4 * p.Cj, cq = p.C >> 2;
#ifdef SNNHIP_IRBW_ABL_NOSTAGE // ablation build (wrong results): no staging of the x tile
        const int total = 0;
#else
        const int total = p.MT * 16 * quads;
#endif
        for (int base = lane; base < total; base += 8 * 64) {
            float4 v[8];
            int lo[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int e = base + r * 64;
                v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                lo[r] = -1;
                if (e < total) {
                    // (divisions by run-time values are ~25 instructions each: two per element made this staging a quarter of the wave's VALU work)
                    // element -> (pixel, quad): EIGHT CONSECUTIVE LANES = eight consecutive pixels of one quad, the next eight lanes the next quad of the same
                    // pixels.  A ds_write_b128 is serviced eight lanes at a time and the planes are 0 mod 256 bytes apart: with the quad running fastest
                    // (lanes 0-7 = the 8 quads of ONE pixel) every store was an 8-way bank conflict -- 72 % of the LDS cycles of MobileNetV2's head block
                    // (PMC), half of its LDS time.  The global side is unchanged: a wave instruction still covers whole pixels.
                    const int blk = static_cast<int>(__umulhi(static_cast<unsigned>(e >> 3), p.magicQuads)), q = (e >> 3) - blk * quads;
                    const int hp = blk * 8 + (e & 7);
                    const int hy = static_cast<int>(__umulhi(static_cast<unsigned>(hp), p.magicHWd)), hx = hp - hy * p.HWd;
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

    [[maybe_unused]] float sxS = 1.0f, sxInv = 1.0f;
    [[maybe_unused]] bool scaled = false;
    if constexpr (S16) { // the wave's x tile in place (its LDS operations complete in order): fp32 x 4 -> [hi x 4 | lo x 4] per 16-byte slot; zeros stay zeros
        static_assert(!S16 || (SNNHIP_IRB_FOLD_BN && R6 && !STEM), "the split form keeps the hidden slice scaled per channel: folded ReLU6 expand epilogue only");
        const int quads = 4 * p.Cj, npx = p.MT * 16;
        float m = 0.f;
        for (int q = 0; q < quads; ++q)
            for (int hp = lane; hp < npx; hp += 64) {
                const float4 v = *reinterpret_cast<const float4*>(xs + q * p.xPlane + hp * 4);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        scaled = split_tile_scale(__builtin_amdgcn_readfirstlane(__float_as_int(m)), sxS, sxInv); // (wave-uniform: the rare scaled path is undone after the expand MFMAs)
        for (int q = 0; q < quads; ++q)
            for (int hp = lane; hp < npx; hp += 64) {
                float4* const sp = reinterpret_cast<float4*>(xs + q * p.xPlane + hp * 4);
                const float4 v = *sp;
                *sp = __builtin_bit_cast(float4, split_f16x3(v.x * sxS, v.y * sxS, v.z * sxS, v.w * sxS));
            }
    }
#ifdef SNNHIP_IRB_TRACE
    if (itrace) istamp[1] = __builtin_readcyclecounter();
#endif
    int hp0[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int pl = g * 16 + n16; // 16 consecutive pixels of the tile in row-major order (two rows of 8)
        hp0[g] = (pl >> 3) * p.s * p.HWd + (pl & 7) * p.s;
#ifdef SNNHIP_IRB_ABL_NOCONF // ablation build (wrong results): 16 consecutive slots per tap read = no bank conflict
        hp0[g] = g * 16 + n16;
#endif
    }
    f32x4 acc[NCBT][G];
#pragma unroll
    for (int cb = 0; cb < NCBT; ++cb)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[cb][g] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < p.nChunks; ++c) {
        const float4* web = weg + static_cast<size_t>(c) * p.wePieces * 64; // this slice's blobs (the LDS images of the block-tile kernel), read in place
        const float4* wpb = wpg + static_cast<size_t>(c) * p.wpPieces * 64;
        // ---- expand: hidden slice c over the wave's halo tile
        if (!p.noExpand) {
            float4 a[CJT];
#pragma unroll
            for (int j = 0; j < CJT; ++j)
                if (j < p.Cj) a[j] = web[j * 64 + lane];
            const float4 sc = web[p.Cj * 64 + k], sh = web[p.Cj * 64 + 4 + k];
            for (int t = 0; t < p.MT; t += 2) {
                const int px0 = t * 16 + n16;
                const bool two = t + 1 < p.MT; // wave-uniform
                const int px1 = two ? px0 + 16 : px0;
#if SNNHIP_IRB_FOLD_BN
                f32x4 acc0 = {sh.x, sh.y, sh.z, sh.w}, acc1 = acc0;
                if (S16 && scaled) acc0 = acc1 = f32x4{sh.x * sxS, sh.y * sxS, sh.z * sxS, sh.w * sxS};
#else
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
                for (int j = 0; j < CJT; ++j)
                    if constexpr (S16) { // (planes past C hold zeros: a 24-channel tile needs no tail form; one MFMA shape per chain, irb_band_kernel's note)
                        if (j < p.Cj) {
                            const f16x8 c0 = __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(xs + (4 * j + k) * p.xPlane + px0 * 4));
                            const f16x8 c1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(xs + (4 * j + k) * p.xPlane + px1 * 4));
                            const f16x8 aj = __builtin_bit_cast(f16x8, a[j]);
                            const f16x8 ahh = __builtin_shufflevector(aj, aj, 0, 1, 2, 3, 0, 1, 2, 3);
                            const f16x8 al0 = __builtin_shufflevector(aj, f16x8{0, 0, 0, 0, 0, 0, 0, 0}, 4, 5, 6, 7, 8, 9, 10, 11);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahh, c0, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahh, c1, acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, c0, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, c1, acc1, 0, 0, 0);
                        }
                    } else if (j == p.Cj - 1 && p.tail8) { // (wave-uniform)
                        const float* const q0 = xs + 4 * j * p.xPlane + px0 * 4 + k;
                        const float* const q1 = xs + 4 * j * p.xPlane + px1 * 4 + k;
                        const float b00 = q0[0], b01 = q0[p.xPlane], b10 = q1[0], b11 = q1[p.xPlane];
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b00, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b10, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b01, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b11, acc1, 0, 0, 0);
                    } else if (j < p.Cj) {
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
                typedef float v2f __attribute__((ext_vector_type(2)));
                [[maybe_unused]] const v2f sc01 = {sc.x, sc.y}, sc23 = {sc.z, sc.w}, sh01 = {sh.x, sh.y}, sh23 = {sh.z, sh.w};
                float4 h;
                {
#if SNNHIP_IRB_FOLD_BN
                    v2f u01 = {acc0[0], acc0[1]}, u23 = {acc0[2], acc0[3]};
                    if (S16 && scaled) { u01 *= sxInv; u23 *= sxInv; }
#else
                    const v2f u01 = __builtin_elementwise_fma(sc01, v2f{acc0[0], acc0[1]}, sh01), u23 = __builtin_elementwise_fma(sc23, v2f{acc0[2], acc0[3]}, sh23);
#endif
                    if constexpr (S16) { // (the slice's channels stay times their weight rows' powers of two: sc holds 6 x that power, the depthwise taps its inverse)
                        h.x = irb_relu_to(u01[0], sc.x);
                        h.y = irb_relu_to(u01[1], sc.y);
                        h.z = irb_relu_to(u23[0], sc.z);
                        h.w = irb_relu_to(u23[1], sc.w);
                    } else {
                        h.x = irb_act<R6>(p.ac1, u01[0]);
                        h.y = irb_act<R6>(p.ac1, u01[1]);
                        h.z = irb_act<R6>(p.ac1, u23[0]);
                        h.w = irb_act<R6>(p.ac1, u23[1]);
                    }
                }
                if (border) {
                    const float m0 = msk[px0];
                    h.x *= m0; h.y *= m0; h.z *= m0; h.w *= m0;
                }
                *reinterpret_cast<float4*>(hs + k * p.hPlane + px0 * 4) = h;
                if (two) {
#if SNNHIP_IRB_FOLD_BN
                    v2f u01 = {acc1[0], acc1[1]}, u23 = {acc1[2], acc1[3]};
                    if (S16 && scaled) { u01 *= sxInv; u23 *= sxInv; }
#else
                    const v2f u01 = __builtin_elementwise_fma(sc01, v2f{acc1[0], acc1[1]}, sh01), u23 = __builtin_elementwise_fma(sc23, v2f{acc1[2], acc1[3]}, sh23);
#endif
                    if constexpr (S16) {
                        h.x = irb_relu_to(u01[0], sc.x);
                        h.y = irb_relu_to(u01[1], sc.y);
                        h.z = irb_relu_to(u23[0], sc.z);
                        h.w = irb_relu_to(u23[1], sc.w);
                    } else {
                        h.x = irb_act<R6>(p.ac1, u01[0]);
                        h.y = irb_act<R6>(p.ac1, u01[1]);
                        h.z = irb_act<R6>(p.ac1, u23[0]);
                        h.w = irb_act<R6>(p.ac1, u23[1]);
                    }
                    if (border) {
                        const float m1 = msk[px1];
                        h.x *= m1; h.y *= m1; h.z *= m1; h.w *= m1;
                    }
                    *reinterpret_cast<float4*>(hs + k * p.hPlane + px1 * 4) = h;
                }
            }
        }
        // ---- depthwise + project of slice c (the wave's LDS operations complete in order: the hidden slice above is visible to all its lanes)
        {
            const float* hb = p.noExpand ? xs + (4 * c + k) * p.xPlane : hs + k * p.hPlane; // (no expand layer: the block input, zero outside the image)
            const float4* dwb = wpb + p.NCB * 64; // [9 taps][16 channels], then scale[16], shift[16]
            float4 wd[9];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) wd[tp] = dwb[tp * 4 + k];
            [[maybe_unused]] const float4 sc = dwb[36 + k];
            const float4 sh = dwb[40 + k];
            float4 dv[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                // the 36 tap FMAs as 18 v_pk_fma_f32 (two channels per instruction: the kernel is bound by VALU issue, and the packed form retires two
                // FMAs per lane in one slot); same products and the same per-channel summation order as the scalar form
                typedef float v2f __attribute__((ext_vector_type(2)));
#if SNNHIP_IRB_FOLD_BN
                v2f s01 = {sh.x, sh.y}, s23 = {sh.z, sh.w};
#else
                v2f s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#endif
#pragma unroll
                for (int fy = 0; fy < 3; ++fy)
#pragma unroll
                    for (int fx = 0; fx < 3; ++fx) {
                        const float4 h = *reinterpret_cast<const float4*>(hb + (hp0[g] + fy * p.HWd + fx) * 4);
                        const float4 w = wd[fy * 3 + fx];
                        s01 = __builtin_elementwise_fma(v2f{h.x, h.y}, v2f{w.x, w.y}, s01);
                        s23 = __builtin_elementwise_fma(v2f{h.z, h.w}, v2f{w.z, w.w}, s23);
                    }
#if SNNHIP_IRB_FOLD_BN
                const v2f t01 = s01, t23 = s23;
#else
                const v2f t01 = __builtin_elementwise_fma(v2f{sc.x, sc.y}, s01, v2f{sh.x, sh.y}), t23 = __builtin_elementwise_fma(v2f{sc.z, sc.w}, s23, v2f{sh.z, sh.w});
#endif
                dv[g].x = irb_act<R6>(p.ac2, t01[0]);
                dv[g].y = irb_act<R6>(p.ac2, t01[1]);
                dv[g].z = irb_act<R6>(p.ac2, t23[0]);
                dv[g].w = irb_act<R6>(p.ac2, t23[1]);
            }
            [[maybe_unused]] f16x8 dhh[G], dl0[G];
            if constexpr (S16) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const f16x8 db = split_f16x3(dv[g].x, dv[g].y, dv[g].z, dv[g].w);
                    dhh[g] = __builtin_shufflevector(db, db, 0, 1, 2, 3, 0, 1, 2, 3);
                    dl0[g] = __builtin_shufflevector(db, f16x8{0, 0, 0, 0, 0, 0, 0, 0}, 4, 5, 6, 7, 8, 9, 10, 11);
                }
            }
#pragma unroll
            for (int cb = 0; cb < NCBT; ++cb)
                if (cb < p.NCB) {
                    const float4 a = wpb[cb * 64 + lane];
                    if constexpr (S16) { // (A = the weight slot [wh | wl]; B = [dh | dh], then [dl | 0]: irb_image_kernel's note)
#pragma unroll
                        for (int g = 0; g < G; ++g) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), dhh[g], acc[cb][g], 0, 0, 0);
#pragma unroll
                        for (int g = 0; g < G; ++g) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), dl0[g], acc[cb][g], 0, 0, 0);
                        continue;
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, dv[g].x, acc[cb][g], 0, 0, 0);
                        acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, dv[g].y, acc[cb][g], 0, 0, 0);
                        acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, dv[g].z, acc[cb][g], 0, 0, 0);
                        acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, dv[g].w, acc[cb][g], 0, 0, 0);
                    }
                }
        }
    }

#ifdef SNNHIP_IRB_TRACE
    if (itrace) istamp[2] = __builtin_readcyclecounter();
#endif
    // ---- epilogue: lane holds output channels 16 cb + 4k .. + 3 of its pixels
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int pl = g * 16 + n16;
        const int oyl = pl >> 3, oxl = pl & 7;
        const int oy = oy0 + oyl, ox = ox0 + oxl;
        if (oy >= p.OH || ox >= p.OW) continue;
        float* yp = y + ((static_cast<size_t>(img) * p.OH + oy) * p.OW + ox) * p.Co;
        const int hpc = (oyl + p.pady) * p.HWd + oxl + p.padx; // residual (stride 1): the block input at the output pixel
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
                if (p.hasRes) { // (the split form's tile is no longer the fp32 input; stride 1, C == Co: the input pixel sits where the output pixel does, in L2)
                    const float4 r = S16 ? *reinterpret_cast<const float4*>(x + ((static_cast<size_t>(img) * p.H + oy) * p.W + ox) * p.C + co)
                                         : *reinterpret_cast<const float4*>(xs + (co >> 2) * p.xPlane + hpc * 4);
                    o.x = apply_act<true>(p.ac4, o.x + r.x, 0.f);
                    o.y = apply_act<true>(p.ac4, o.y + r.y, 0.f);
                    o.z = apply_act<true>(p.ac4, o.z + r.z, 0.f);
                    o.w = apply_act<true>(p.ac4, o.w + r.w, 0.f);
                }
                *reinterpret_cast<float4*>(yp + co) = o;
            }
        }
    }
#ifdef SNNHIP_IRB_TRACE
    if (itrace)
        printf("irbtrace G%d NCB%d Cj%d chunks %d noexp %d wave %d: stage %llu slices %llu (%llu each) epi %llu (stem patch %llu)\n", G, p.NCB, p.Cj, p.nChunks, p.noExpand, wave,
               istamp[1] - istamp[0], istamp[2] - istamp[1], (istamp[2] - istamp[1]) / p.nChunks, __builtin_readcyclecounter() - istamp[2], STEM ? istamp[3] - istamp[0] : 0ull);
#endif
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
        if (p.stemK) SNNHIP_REQUIRE(x->n == p.N && x->h == p.IH && x->w == p.IW && x->c == 3, "irb (stem): input dims %dx%dx%dx%d != plan %dx%dx%dx3", x->n, x->h, x->w, x->c, p.N, p.IH, p.IW);
        else
            SNNHIP_REQUIRE(x->n == p.N && x->h == p.H && x->w == p.W && x->c == p.C, "irb: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h, x->w, x->c,
                           p.N, p.H, p.W, p.C);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.Co, "irb: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n,
                       out->h, out->w, out->c, p.N, p.OH, p.OW, p.Co);
        SNNHIP_LAUNCH(kernel, grid, dim3(static_cast<unsigned>(threads)), ldsBytes, ctx->stream, p, x->data, reinterpret_cast<const float4*>(d_we),
                           reinterpret_cast<const float4*>(d_wp), reinterpret_cast<const float4*>(d_e3), out->data);
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

// ---- whole-image form (round 4): the 14x14 / 7x7 blocks of MobileNetV2 (b07-b16).  On these maps the tile-per-wave kernel above loses to the three
// separate layers (a 4x8 tile of a 14x14 image is mostly halo, and every wave walks all 24-60 weight slices), and the separate layers run at 0.22-0.27
// of the MFMA roofline: expand 32 us + depthwise 41 us + project/add 46 us for 64 -> 384 -> 64 at batch 256, 31 us of matrix work.  Here a BLOCK owns
// one whole IMAGE -- no halo at all -- and each of its four waves (one per SIMD) a QUARTER OF THE HIDDEN CHANNELS (slices c = wave, wave + 4, ...):
//   x tile     the image, [pixel][C / 4 + 1] 16-byte slots (odd pitch: conflict-free operand reads), copied once by LDS-DMA; also the residual
//   E(c)       expand MFMAs over the image's ceil(HW / 16) pixel tiles (B operand straight from the x tile) -> act -> the wave's PRIVATE hidden slice,
//              a zero-bordered (H + pad) x (W + pad) tile of 4 quad planes: the depthwise layer's zero padding is the border, written once
//   D(c), P(c) depthwise taps per output pixel tile -> act -> the B operand of the project MFMAs, accumulated in the wave's own acc[Co / 16][tiles]
//              (a wave's hidden channels are read by no other wave: NO barrier in the loop; every weight is read once per image)
//   reduce     the four waves' partial sums over their hidden quarters meet in LDS, one 16-channel output block at a time (fixed order), then
//              scale / shift / act [+ x -> act] and 16-byte stores.
// One wave per SIMD with up to 320 accumulator registers: latency is hidden inside the wave (next slice's weights requested a phase ahead, independent
// MFMA chains), not by occupancy.  Weight blobs: the slice format of irb_wave_kernel.
struct IrbImgParams {
    int N, H, W, C, Ch, Co, OH, OW, s;
    int HW, OHW;
    int MT;            // ceil(HW / 16) expand MFMA tiles
    int SP;            // 16-byte slots per pixel of the x tile: C / 4 + 1
    int HWd, hPlane4;  // hidden tile: row pitch in pixels, float4 per quad plane (>= rows * HWd + 1: the last slot takes the results of padding pixels)
    int offH4;         // float4 offset of the hidden region (4 waves x 4 planes); the pixel tables follow it
    int slicesPerWave; // Ch / 64
    int wePieces, wpPieces;
    int hasRes;
    int OS;            // blocks per image (round 6): block b computes output blocks (b % OS) * NCB .. of image b / OS -- MobileNetV2's b16 (160 -> 960 -> 320: 20 output blocks,
                       // 320 accumulator registers per lane) as two blocks of ten, each with its own expand / depthwise pass
    unsigned magicSP;
    ActCfg ac1, ac2, ac3, ac4;
};

template <int NCB /* Co / 16 / OS */, int CJ /* C / 16 */, int G /* output pixel tiles */, bool R6, bool S16 = false /* split-precision pointwise stages */>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void irb_image_kernel(IrbImgParams p, const float* __restrict__ x, const float4* __restrict__ weg,
                                                                                                 const float4* __restrict__ wpg, const float4* __restrict__ epi3,
                                                                                                 const int* __restrict__ tabs, float* __restrict__ y) {
    extern __shared__ float4 sm4[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n16 = lane & 15, k = lane >> 4;
    const int img = p.OS > 1 ? blockIdx.x / p.OS : blockIdx.x;
    const int cb0 = p.OS > 1 ? (blockIdx.x % p.OS) * NCB : 0, ncbAll = NCB * p.OS; // this block's first output block; the blobs hold all of them
#ifdef SNNHIP_IRBI_TRACE // experiment builds (tools/exp_one.sh): one block prints the s_memtime stamps of its phases
    const bool itr = blockIdx.x == 100 && lane == 0 && (wave == 0 || wave == 3);
    unsigned long long ist[8] = {};
    if (itr) ist[0] = __builtin_readcyclecounter();
#define IRBI_MARK(i) do { if (itr) ist[i] = __builtin_readcyclecounter(); } while (0)
#else
#define IRBI_MARK(i) do { } while (0)
#endif
#if defined(SNNHIP_IRBI_TRACE) && SNNHIP_IRBI_TRACE >= 2
    const unsigned long long censusT0 = wall_clock64();
    unsigned long long censusT1 = 0;
#endif
    float4* const xs4 = sm4;
    float4* const hs4 = sm4 + p.offH4 + wave * 4 * p.hPlane4;
    int* const tabE = reinterpret_cast<int*>(sm4 + p.offH4 + 16 * p.hPlane4); // [MT * 16]: hidden position of x-tile pixel i
    int* const tabD = tabE + p.MT * 16;                                        // [G * 16]: hidden position of output pixel o's first tap
    {
        const float* xi = x + static_cast<size_t>(img) * p.HW * p.C;
        const int totalSlots = p.HW * p.SP;
        const unsigned xsLds = lds_byte_addr(xs4);
        for (int e0 = wave * 64; e0 < totalSlots; e0 += 256) { // (wave-uniform)
            const int e = e0 + lane;
            const int px = static_cast<int>(__umulhi(static_cast<unsigned>(e), p.magicSP)), sl = e - px * p.SP;
            // the pad slot of a pixel and the slots past the image re-read the image's first bytes: nothing reads what they write
            lds_dma16_sbase(xi, (e < totalSlots && sl < p.SP - 1) ? static_cast<unsigned>(px * p.C + sl * 4) * 4u : 0u, xsLds + static_cast<unsigned>(e0) * 16u);
        }
        for (int i = lane; i < 4 * p.hPlane4; i += 64) hs4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = tid; i < (p.MT + G) * 16; i += 256) tabE[i] = tabs[i];
        if (S16 && tid == 0) tabD[G * 16] = 0; // the image's largest magnitude (bit pattern of a non-negative float)
        lds_dma_wait();
    }
    __syncthreads();
    [[maybe_unused]] float sxInv = 1.0f;
    if constexpr (S16) { // the x tile in place: fp32 x 4 -> [hi x 4 | lo x 4] per 16-byte slot, scaled by a power of two if the image reaches 2^15
        const int totalSlots = p.HW * p.SP;
        float m = 0.f;
        for (int e = tid; e < totalSlots; e += 256) {
            const float4 v = xs4[e];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) atomicMax(&tabD[G * 16], __float_as_int(m));
        __syncthreads();
        float sx = 1.0f;
        (void)split_tile_scale(tabD[G * 16], sx, sxInv);
        for (int e = tid; e < totalSlots; e += 256) {
            const float4 v = xs4[e];
            xs4[e] = __builtin_bit_cast(float4, split_f16x3(v.x * sx, v.y * sx, v.z * sx, v.w * sx));
        }
        __syncthreads();
    }
    IRBI_MARK(1);
#if defined(SNNHIP_IRBI_TRACE) && SNNHIP_IRBI_TRACE >= 2
    censusT1 = wall_clock64(); // staging done
#endif

    f32x4 acc[NCB][G];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[cb][g] = f32x4{0.f, 0.f, 0.f, 0.f};

    const size_t weStep = static_cast<size_t>(4) * p.wePieces * 64, wpStep = static_cast<size_t>(4) * p.wpPieces * 64;
    const float4* web = weg + static_cast<size_t>(wave) * p.wePieces * 64; // slices wave, wave + 4, ...
    const float4* wpb = wpg + static_cast<size_t>(wave) * p.wpPieces * 64;
    float4 a[CJ], ap[NCB], wd[9];
#pragma unroll
    for (int j = 0; j < CJ; ++j) a[j] = web[j * 64 + lane];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) ap[cb] = wpb[(cb0 + cb) * 64 + lane];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) wd[tp] = wpb[ncbAll * 64 + tp * 4 + k];
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int eFirst0 = tabE[n16], eFirst1 = tabE[(p.MT > 1 ? 16 : 0) + n16];
#ifdef SNNHIP_IRBI_ABL
    float ablSink = 0.f;
#endif

    for (int i = 0; i < p.slicesPerWave; ++i) {
        const bool more = i + 1 < p.slicesPerWave; // (wave-uniform)
        if (i == 1) IRBI_MARK(2);
        if (i == 2) IRBI_MARK(4);
        // (the epilogue constants of a phase are requested at its start -- their first use is a tile's worth of MFMAs / tap FMAs away -- instead of with the
        // weights a phase earlier: 16 registers the 96 -> 576 -> 96 instantiation does not have)
        float4 sc1 = web[CJ * 64 + k];
        const float4 sh1 = web[CJ * 64 + 4 + k];
        if constexpr (S16) { // (the power of two the image was scaled by)
            static_assert(!S16 || !SNNHIP_IRBI_FOLD_BN, "the split form undoes the input scale in the expand epilogue's multiplier");
            sc1.x *= sxInv; sc1.y *= sxInv; sc1.z *= sxInv; sc1.w *= sxInv;
        }
        // ---- E: the wave's hidden slice over the whole image, two pixel tiles in flight
        int eNext0 = eFirst0, eNext1 = eFirst1;
        for (int t = 0; t < p.MT; t += 2) {
            const int px0 = t * 16 + n16;
            const bool two = t + 1 < p.MT;
            const int px1 = two ? px0 + 16 : px0;
            // (round 6) this pair's hidden positions were read a pair ahead (the first pair's once per block): one wave per SIMD, nobody hides the look-up
            const int e0 = eNext0, e1 = eNext1;
            if (t + 2 < p.MT) {
                eNext0 = tabE[px0 + 32];
                eNext1 = tabE[t + 3 < p.MT ? px0 + 48 : px0 + 32];
            }
            const float4* const b0p = xs4 + px0 * p.SP + k;
            const float4* const b1p = xs4 + px1 * p.SP + k;
#if SNNHIP_IRBI_FOLD_BN
            f32x4 acc0 = {sh1.x, sh1.y, sh1.z, sh1.w}, acc1 = acc0;
#else
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#endif
            [[maybe_unused]] f32x4 lo0 = {0.f, 0.f, 0.f, 0.f}, lo1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                const float4 b0 = b0p[4 * j], b1 = b1p[4 * j];
                if constexpr (S16) {
                    const f16x8 aj = __builtin_bit_cast(f16x8, a[j]), c0 = __builtin_bit_cast(f16x8, b0), c1 = __builtin_bit_cast(f16x8, b1);
                    // the two MFMA shapes keep their OWN accumulator chains (an x16 reading what an x32 has just written is a hazard the compiler does not pad), summed
                    // in the epilogue: A = [wh | wh] against the slot [xh | xl], A = wl against its lower half (the all-x32 form of irb_band_kernel spills here)
                    const f16x8 ahh = __builtin_shufflevector(aj, aj, 0, 1, 2, 3, 0, 1, 2, 3);
                    const f16x4 al = __builtin_shufflevector(aj, aj, 4, 5, 6, 7);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahh, c0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahh, c1, acc1, 0, 0, 0);
                    lo0 = __builtin_amdgcn_mfma_f32_16x16x16f16(al, __builtin_shufflevector(c0, c0, 0, 1, 2, 3), lo0, 0, 0, 0);
                    lo1 = __builtin_amdgcn_mfma_f32_16x16x16f16(al, __builtin_shufflevector(c1, c1, 0, 1, 2, 3), lo1, 0, 0, 0);
                    continue;
                }
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b1.w, acc1, 0, 0, 0);
            }
            if constexpr (S16) {
                acc0 += lo0;
                acc1 += lo1;
            }
#if defined(SNNHIP_IRBI_ABL) && (SNNHIP_IRBI_ABL & 1) // ablation build (tools/r6_iabl.sh): E without its epilogue (scale / ReLU6 / LDS write)
            ablSink += acc0[0] + acc1[0] + static_cast<float>(e0 + e1);
            continue;
#endif
            [[maybe_unused]] const v2f sc01 = {sc1.x, sc1.y}, sc23 = {sc1.z, sc1.w}, sh01 = {sh1.x, sh1.y}, sh23 = {sh1.z, sh1.w};
            {
#if SNNHIP_IRBI_FOLD_BN
                const v2f u01 = {acc0[0], acc0[1]}, u23 = {acc0[2], acc0[3]};
#else
                const v2f u01 = __builtin_elementwise_fma(sc01, v2f{acc0[0], acc0[1]}, sh01), u23 = __builtin_elementwise_fma(sc23, v2f{acc0[2], acc0[3]}, sh23);
#endif
                hs4[k * p.hPlane4 + e0] = make_float4(irb_act<R6>(p.ac1, u01[0]), irb_act<R6>(p.ac1, u01[1]), irb_act<R6>(p.ac1, u23[0]), irb_act<R6>(p.ac1, u23[1]));
            }
            if (two) {
#if SNNHIP_IRBI_FOLD_BN
                const v2f u01 = {acc1[0], acc1[1]}, u23 = {acc1[2], acc1[3]};
#else
                const v2f u01 = __builtin_elementwise_fma(sc01, v2f{acc1[0], acc1[1]}, sh01), u23 = __builtin_elementwise_fma(sc23, v2f{acc1[2], acc1[3]}, sh23);
#endif
                hs4[k * p.hPlane4 + e1] = make_float4(irb_act<R6>(p.ac1, u01[0]), irb_act<R6>(p.ac1, u01[1]), irb_act<R6>(p.ac1, u23[0]), irb_act<R6>(p.ac1, u23[1]));
            }
        }
        if (i == 1) IRBI_MARK(3);
        // the next slice's expand weights are requested now and arrive under D / P (one wave per SIMD: nobody else hides the L2 round trip)
        __builtin_amdgcn_sched_barrier(0);
        if (more) web += weStep;
#pragma unroll
        for (int j = 0; j < CJ; ++j) a[j] = web[j * 64 + lane];
        [[maybe_unused]] const float4 sc2 = wpb[ncbAll * 64 + 36 + k];
        const float4 sh2 = wpb[ncbAll * 64 + 40 + k];
        __builtin_amdgcn_sched_barrier(0);
        // ---- D + P: depthwise taps of output tile g -> the B operand of the project MFMAs (kk outer, cb inner: consecutive MFMAs are independent).
        // The order is pinned (sched_barrier): tile g's tap FMAs, then the 9 tap reads of tile g + 1 INTO THE SAME REGISTERS, then tile g's MFMAs, under
        // which they arrive.  Left alone the scheduler hoists the tap reads of all 13 tiles to the top (468 registers next to 208-312 accumulators: scratch).
        // (Software-pipelined by one tile -- tile g's MFMAs interleaved by sched_group_barrier with tile g + 1's packed tap FMAs and the refills of the tap
        // registers -- it ran 2-6 % SLOWER on every block (b07 69.3 -> 73.7 us): fp32 VALU work beside fp32 MFMAs is not hidden, it shares their issue port.)
        const float4* const hb = hs4 + k * p.hPlane4;
        float4 h[9];
        // (round 6) the table entry of tile g + 1 is read a tile ahead, behind the tap reads of tile g: the nine reads of a tile used to start with an LDS
        // round trip for their own base address (the ablation that replaced the look-up by n16 ran b07 68.6 -> 65.0 us: profiles/r06_irb_*)
        int hpNext = 0;
        {
#ifdef SNNHIP_IRB_ABL_NOCONF
            const int hp0 = n16;
#else
            const int hp0 = tabD[n16];
#endif
#if defined(SNNHIP_IRBI_ABL) && (SNNHIP_IRBI_ABL & 2) // ablation build: D without its tap reads and FMAs
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) h[tp] = wd[tp];
            ablSink += static_cast<float>(hp0 + hpNext);
#else
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) h[tp] = hb[hp0 + (tp / 3) * p.HWd + tp % 3];
            if (G > 1) hpNext = tabD[16 + n16];
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
#if defined(SNNHIP_IRBI_ABL) && (SNNHIP_IRBI_ABL & 2)
            const float d0 = sc2.x, d1 = sc2.y, d2 = sh2.x, d3 = sh2.y;
            if (false) {
#else
#if SNNHIP_IRBI_FOLD_BN
            v2f s01 = {sh2.x, sh2.y}, s23 = {sh2.z, sh2.w};
#else
            v2f s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#endif
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                s01 = __builtin_elementwise_fma(v2f{h[tp].x, h[tp].y}, v2f{wd[tp].x, wd[tp].y}, s01);
                s23 = __builtin_elementwise_fma(v2f{h[tp].z, h[tp].w}, v2f{wd[tp].z, wd[tp].w}, s23);
            }
#if SNNHIP_IRBI_FOLD_BN
            const v2f t01 = s01, t23 = s23;
#else
            const v2f t01 = __builtin_elementwise_fma(v2f{sc2.x, sc2.y}, s01, v2f{sh2.x, sh2.y}), t23 = __builtin_elementwise_fma(v2f{sc2.z, sc2.w}, s23, v2f{sh2.z, sh2.w});
#endif
            const float d0 = irb_act<R6>(p.ac2, t01[0]), d1 = irb_act<R6>(p.ac2, t01[1]), d2 = irb_act<R6>(p.ac2, t23[0]), d3 = irb_act<R6>(p.ac2, t23[1]);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < G) {
#endif
#ifdef SNNHIP_IRB_ABL_NOCONF
                const int hp0 = ((g + 1) & 7) * 16 + n16;
#else
                const int hp0 = hpNext;
#endif
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) h[tp] = hb[hp0 + (tp / 3) * p.HWd + tp % 3];
                if (g + 2 < G) hpNext = tabD[(g + 2) * 16 + n16];
            }
            if constexpr (S16) {
                // ONE MFMA shape on the output accumulators (an x16 reading what an x32 has just written is a hazard the compiler does not pad): the weight slot
                // [wh | wl] is the A operand of both products as it is, B = [dh | dh] gives wh dh + wl dh, B = [dl | 0] gives wh dl; both built once per tile
                const f16x8 db = split_f16x3(d0, d1, d2, d3);
                const f16x8 dhh = __builtin_shufflevector(db, db, 0, 1, 2, 3, 0, 1, 2, 3);
                const f16x8 dl0 = __builtin_shufflevector(db, f16x8{0, 0, 0, 0, 0, 0, 0, 0}, 4, 5, 6, 7, 8, 9, 10, 11);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ap[cb]), dhh, acc[cb][g], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ap[cb]), dl0, acc[cb][g], 0, 0, 0);
            } else {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[cb].x, d0, acc[cb][g], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[cb].y, d1, acc[cb][g], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[cb].z, d2, acc[cb][g], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[cb].w, d3, acc[cb][g], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) wpb += wpStep;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) ap[cb] = wpb[(cb0 + cb) * 64 + lane];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) wd[tp] = wpb[ncbAll * 64 + tp * 4 + k];
    }

#ifdef SNNHIP_IRBI_ABL
    if (p.N < 0) y[tid] = ablSink;
#endif
    // ---- the four hidden quarters meet, one 16-channel output block at a time: [wave][tile][lane] partial sums in the (now free) hidden region
    IRBI_MARK(5);
    __syncthreads();
    IRBI_MARK(6);
    float4* const red4 = sm4 + p.offH4;
    float* const yi = y + static_cast<size_t>(img) * p.OHW * p.Co;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int g = 0; g < G; ++g) red4[(wave * G + g) * 64 + lane] = make_float4(acc[cb][g][0], acc[cb][g][1], acc[cb][g][2], acc[cb][g][3]);
        __syncthreads();
        const float4 sc = epi3[2 * ((cb0 + cb) * 4 + k)], sh = epi3[2 * ((cb0 + cb) * 4 + k) + 1];
        for (int g = wave; g < G; g += 4) {
            const float4 q0 = red4[(0 * G + g) * 64 + lane], q1 = red4[(1 * G + g) * 64 + lane], q2 = red4[(2 * G + g) * 64 + lane], q3 = red4[(3 * G + g) * 64 + lane];
            const int o = g * 16 + n16;
            if (o < p.OHW) {
                float4 r;
                r.x = apply_act<true>(p.ac3, fmaf(sc.x, ((q0.x + q1.x) + q2.x) + q3.x, sh.x), 0.f);
                r.y = apply_act<true>(p.ac3, fmaf(sc.y, ((q0.y + q1.y) + q2.y) + q3.y, sh.y), 0.f);
                r.z = apply_act<true>(p.ac3, fmaf(sc.z, ((q0.z + q1.z) + q2.z) + q3.z, sh.z), 0.f);
                r.w = apply_act<true>(p.ac3, fmaf(sc.w, ((q0.w + q1.w) + q2.w) + q3.w, sh.w), 0.f);
                if (p.hasRes) { // (stride 1, C == Co: output pixel o is x-tile pixel o; the split form's tile is no longer the fp32 input: L2 has it)
                    const float4 xr = S16 ? *reinterpret_cast<const float4*>(x + (static_cast<size_t>(img) * p.HW + o) * p.C + (cb0 + cb) * 16 + 4 * k) : xs4[o * p.SP + (cb0 + cb) * 4 + k];
                    r.x = apply_act<true>(p.ac4, r.x + xr.x, 0.f);
                    r.y = apply_act<true>(p.ac4, r.y + xr.y, 0.f);
                    r.z = apply_act<true>(p.ac4, r.z + xr.z, 0.f);
                    r.w = apply_act<true>(p.ac4, r.w + xr.w, 0.f);
                }
                *reinterpret_cast<float4*>(yi + static_cast<size_t>(o) * p.Co + (cb0 + cb) * 16 + 4 * k) = r;
            }
        }
        if (cb + 1 < NCB) __syncthreads();
    }
#if defined(SNNHIP_IRBI_TRACE) && SNNHIP_IRBI_TRACE >= 2 // census build: every block's wall-clock span (100 MHz ticks) and its XCC / CU
    if (tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        printf("irbc %d xcc %u se %u cu %u : %llu %llu %llu\n", blockIdx.x, xcc & 15, (hw >> 13) & 7, (hw >> 8) & 15, censusT0, censusT1, static_cast<unsigned long long>(wall_clock64()));
    }
#endif
#ifdef SNNHIP_IRBI_TRACE
    if (itr)
        printf("irbi NCB%d CJ%d G%d wave %d: stage %llu | slice 1: E %llu D+P %llu | loop %llu (%d slices) | wait-others %llu reduce+store %llu | total %llu\n", NCB, CJ, G, wave, ist[1] - ist[0],
               ist[3] - ist[2], ist[4] - ist[3], ist[5] - ist[1], p.slicesPerWave, ist[6] - ist[5], __builtin_readcyclecounter() - ist[6], __builtin_readcyclecounter() - ist[0]);
#endif
#undef IRBI_MARK
}

struct IrbImagePlan : snnhip_plan {
    IrbImgParams p;
    float* d_we = nullptr;
    float* d_wp = nullptr;
    float* d_e3 = nullptr;
    float* d_tabs = nullptr;
    size_t ldsBytes = 0;
    void (*kernel)(IrbImgParams, const float*, const float4*, const float4*, const float4*, const int*, float*) = nullptr;

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "inverted-residual block: expects 1 input (the block input is also the residual), got %d", nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.H && x->w == p.W && x->c == p.C, "irb: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h, x->w, x->c, p.N, p.H, p.W, p.C);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.Co, "irb: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n, out->h, out->w,
                       out->c, p.N, p.OH, p.OW, p.Co);
        SNNHIP_LAUNCH(kernel, dim3(static_cast<unsigned>(p.N * p.OS)), dim3(256), ldsBytes, ctx->stream, p, x->data, reinterpret_cast<const float4*>(d_we),
                      reinterpret_cast<const float4*>(d_wp), reinterpret_cast<const float4*>(d_e3), reinterpret_cast<const int*>(d_tabs), out->data);
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

typedef void (*IrbImgFn)(IrbImgParams, const float*, const float4*, const float4*, const float4*, const int*, float*);
IrbImgFn pick_irb_image(int ncb, int cj, int g, bool s16, int* gt) {
#define SNNHIP_IRBI(NCB_, CJ_, G_) \
    if (ncb == NCB_ && cj == CJ_ && g <= G_) return *gt = G_, s16 ? irb_image_kernel<NCB_, CJ_, G_, true, true> : irb_image_kernel<NCB_, CJ_, G_, true, false>;
    SNNHIP_IRBI(4, 4, 13)   // 64 -> 384 -> 64 at 14x14 (MobileNetV2 b07-b09)
    SNNHIP_IRBI(6, 4, 13)   // 64 -> 384 -> 96 (b10)
    SNNHIP_IRBI(6, 6, 13)   // 96 -> 576 -> 96 (b11, b12)
    SNNHIP_IRBI(10, 6, 4)   // 96 -> 576 -> 160, 14x14 -> 7x7 (b13)
    SNNHIP_IRBI(10, 10, 4)  // 160 -> 960 -> 160 at 7x7 (b14, b15)
#undef SNNHIP_IRBI
    return nullptr;
}

// ---- band form (round 4): the blocks on 28x28 .. 112x112 maps (MobileNetV2 b01-b06), where irb_wave_kernel's 2x8 / 4x8 tiles per wave pay 1.5-2x the
// expand layer in halo pixels, pad 24 input channels to 32, and every wave walks every weight slice for 16-32 output pixels.  A BLOCK owns a band of R
// output rows x SW columns of one image:
//   x tile     the band's input pixels (halo rows / columns included, clipped to the image), [pixel][C / 4 + 1] slots, one LDS-DMA pass
//   per slice  E: the block's NW waves share the x tile's pixel tiles (tile t -> wave t mod NW), results into one of TWO hidden buffers (zero-bordered
//              band tile, 4 quad planes) -- one barrier -- D + P: every wave its own output pixel tiles, all output channels: acc[Co / 16][GW].
//              E(c + 1) writes the other buffer, so ONE barrier per slice orders everything.
//   C % 16 == 8 (24 channels): the last 8 channels are TWO MFMAs with 4-byte operand reads (channel 16 j + kk, 16 j + 4 + kk) instead of four half-empty ones.
// Halo work of the expand layer: 1.14-1.5x (rows only, or rows and a strip's two columns) instead of 1.5-2x; the accumulators are small (8-32 registers),
// so 7-8 waves per block (2 per SIMD) hide each other's latencies.
struct IrbBandParams {
    int N, H, W, C, Ch, Co, OH, OW, s, padx, pady;
    int R, SW, nBy, nBx;
    int HWd, hPlane4, SP;
    int offH4, offTab4; // float4 offsets: the two hidden buffers (2 x 4 planes), the tables
    int MTmax;          // pixel tiles of the largest x tile
    int NW, nSlices, wePieces, wpPieces;
    int hasRes, tail8;
    unsigned magicSP;
    ActCfg ac1, ac2, ac3, ac4;
};

template <int NCB, int CJ, int GW, bool R6, bool S16 = false /* split-precision pointwise stages: irb_image_kernel's note */>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NCB <= 2 ? 4 : 2))) void irb_band_kernel(IrbBandParams p, const float* __restrict__ x, const float4* __restrict__ weg, const float4* __restrict__ wpg,
                                                      const float4* __restrict__ epi3, float* __restrict__ y) {
    extern __shared__ float4 sm4[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n16 = lane & 15, k = lane >> 4;
    const int NT = p.NW * 64, OT16 = p.NW * GW * 16;
#ifdef SNNHIP_IRBB_TRACE // experiment builds (tools/exp_one.sh): one block sums the s_memtime spans of its phases over the slices and prints them
    const bool btr = blockIdx.x == 1500 && lane == 0 && (wave == 0 || wave == p.NW - 1);
    unsigned long long bT = __builtin_readcyclecounter(), bAcc[6] = {};
#define IRBB_ADD(i) do { if (btr) { const unsigned long long n_ = __builtin_readcyclecounter(); bAcc[i] += n_ - bT; bT = n_; } } while (0)
#else
#define IRBB_ADD(i) do { } while (0)
#endif
    const int bx = blockIdx.x % p.nBx, by = (blockIdx.x / p.nBx) % p.nBy, img = blockIdx.x / (p.nBx * p.nBy);
    // band geometry (block-uniform)
    const int oy0 = by * p.R, ox0 = bx * p.SW;
    const int Rb = min(p.R, p.OH - oy0), OWb = min(p.SW, p.OW - ox0);
    const int hy0 = oy0 * p.s - p.pady, hx0 = ox0 * p.s - p.padx;
    const int iyA = max(hy0, 0), iyB = min(hy0 + (Rb - 1) * p.s + 3, p.H), ixA = max(hx0, 0), ixB = min(hx0 + (OWb - 1) * p.s + 3, p.W);
    const int Wx = ixB - ixA, nPx = (iyB - iyA) * Wx, MT = (nPx + 15) >> 4;
    const unsigned magicWx = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(Wx) - 1) / static_cast<unsigned>(Wx));
    float4* const xs4 = sm4;
    float4* const hsb = sm4 + p.offH4;
    int* const tabE = reinterpret_cast<int*>(sm4 + p.offTab4); // [MTmax * 16] hidden position of x-tile pixel i (padding pixels: the plane's last slot)
    int* const tabD = tabE + p.MTmax * 16;                     // [OT16] hidden position of output pixel o's first tap
    int* const tabO = tabD + OT16;                             // [OT16] float offset of output pixel o in the image's output, -1 outside the band
    int* const tabR = tabO + OT16;                             // [OT16] x-tile pixel of output pixel o (the residual)
    {
        for (int i = tid; i < p.MTmax * 16; i += NT) {
            const int row = static_cast<int>(__umulhi(static_cast<unsigned>(i), magicWx));
            tabE[i] = i < nPx ? (iyA - hy0 + row) * p.HWd + (ixA - hx0) + (i - row * Wx) : p.hPlane4 - 1;
        }
        const unsigned magicOWb = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(OWb) - 1) / static_cast<unsigned>(OWb));
        for (int o = tid; o < OT16; o += NT) {
            const int orow = static_cast<int>(__umulhi(static_cast<unsigned>(o), magicOWb)), ocol = o - orow * OWb;
            const bool ok = orow < Rb;
            tabD[o] = ok ? orow * p.s * p.HWd + ocol * p.s : 0;
            tabO[o] = ok ? ((oy0 + orow) * p.OW + ox0 + ocol) * p.Co : -1;
            tabR[o] = ok ? (oy0 + orow - iyA) * Wx + (ox0 + ocol - ixA) : 0;
        }
        for (int i = tid; i < 8 * p.hPlane4; i += NT) hsb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* xi = x + static_cast<size_t>(img) * p.H * p.W * p.C;
        const int totalSlots = nPx * p.SP;
        const unsigned xsLds = lds_byte_addr(xs4);
        const unsigned firstOff = static_cast<unsigned>((iyA * p.W + ixA) * p.C) * 4u; // (pad slots and slots past the tile re-read the tile's first bytes)
        for (int e0 = wave * 64; e0 < totalSlots; e0 += NT) {
            const int e = e0 + lane;
            const int px = static_cast<int>(__umulhi(static_cast<unsigned>(e), p.magicSP)), sl = e - px * p.SP;
            const int row = static_cast<int>(__umulhi(static_cast<unsigned>(px), magicWx)), col = px - row * Wx;
            const unsigned off = static_cast<unsigned>(((iyA + row) * p.W + ixA + col) * p.C + sl * 4) * 4u;
            lds_dma16_sbase(xi, (e < totalSlots && sl < p.SP - 1) ? off : firstOff, xsLds + static_cast<unsigned>(e0) * 16u);
        }
        if (S16 && tid == 0) tabR[OT16] = 0; // the tile's largest magnitude (bit pattern of a non-negative float)
        lds_dma_wait();
    }
    __syncthreads();
    [[maybe_unused]] float sxS = 1.0f, sxInv = 1.0f;
    [[maybe_unused]] bool scaled = false;
    if constexpr (S16) { // the x tile in place: fp32 x 4 -> [hi x 4 | lo x 4] per 16-byte slot (pad slots too: a 24-channel tile's last K group reads them against zero weights)
        static_assert(!S16 || SNNHIP_IRB_FOLD_BN, "the split form of the band kernel keeps the hidden slice scaled per channel: folded expand epilogue only");
        const int totalSlots = nPx * p.SP;
        float m = 0.f;
        for (int e = tid; e < totalSlots; e += NT) {
            const float4 v = xs4[e];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) atomicMax(&tabR[OT16], __float_as_int(m));
        __syncthreads();
        scaled = split_tile_scale(__builtin_amdgcn_readfirstlane(tabR[OT16]), sxS, sxInv); // (block-uniform: the rare scaled path, undone after the expand MFMAs, is a scalar branch)
        for (int e = tid; e < totalSlots; e += NT) {
            const float4 v = xs4[e];
            xs4[e] = __builtin_bit_cast(float4, split_f16x3(v.x * sxS, v.y * sxS, v.z * sxS, v.w * sxS));
        }
        __syncthreads();
    }
    IRBB_ADD(0); // setup: tables, zero fill, x tile DMA, barrier

    f32x4 acc[NCB][GW];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int g = 0; g < GW; ++g) acc[cb][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float4* web = weg;
    const float4* wpb = wpg;
    float4 a[CJ], ap[NCB], wd[9];
#pragma unroll
    for (int j = 0; j < CJ; ++j) a[j] = web[j * 64 + lane];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) ap[cb] = wpb[cb * 64 + lane];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) wd[tp] = wpb[NCB * 64 + tp * 4 + k];
    typedef float v2f __attribute__((ext_vector_type(2)));
    const float* const xsf = reinterpret_cast<const float*>(xs4);
    // (round 6) table entries that are the same in every slice live in registers: the first tile pair's hidden positions, the output tiles' first-tap positions
    const int eFirst0 = tabE[min(wave, MT - 1) * 16 + n16], eFirst1 = tabE[(wave + p.NW < MT ? wave + p.NW : min(wave, MT - 1)) * 16 + n16];
    int hpD[GW];
#pragma unroll
    for (int g = 0; g < GW; ++g) hpD[g] = tabD[(wave + p.NW * g) * 16 + n16];

    for (int c = 0; c < p.nSlices; ++c) {
        const bool more = c + 1 < p.nSlices;
        float4* const hs4 = hsb + (c & 1) * 4 * p.hPlane4;
        [[maybe_unused]] const float4 sc1 = web[CJ * 64 + k]; // (split form: 6 x the power of two the slice's channels are kept at)
        const float4 sh1 = web[CJ * 64 + 4 + k];
        [[maybe_unused]] const float4 sc2 = wpb[NCB * 64 + 36 + k];
        const float4 sh2 = wpb[NCB * 64 + 40 + k];
        // ---- E: this wave's share of the x tile's pixel tiles, two in flight
        int eNext0 = eFirst0, eNext1 = eFirst1;
        for (int t = wave; t < MT; t += 2 * p.NW) {
            const int px0 = t * 16 + n16;
            const bool two = t + p.NW < MT;
            const int px1 = two ? px0 + 16 * p.NW : px0;
            // (round 6) the hidden positions of this pair were read a pair ahead (the first pair's once per block): the look-up used to sit between the
            // MFMAs' results and the LDS write, an exposed round trip per pair
            const int e0 = eNext0, e1 = eNext1;
            if (t + 2 * p.NW < MT) {
                const int q0 = px0 + 32 * p.NW;
                eNext0 = tabE[q0];
                eNext1 = tabE[t + 3 * p.NW < MT ? q0 + 16 * p.NW : q0];
            }
#if SNNHIP_IRB_FOLD_BN
            f32x4 acc0 = {sh1.x, sh1.y, sh1.z, sh1.w}, acc1 = acc0;
            if (S16 && scaled) acc0 = acc1 = f32x4{sh1.x * sxS, sh1.y * sxS, sh1.z * sxS, sh1.w * sxS};
#else
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                if constexpr (S16) {
                    // (a 24-channel tile's last K group: lanes k = 2, 3 re-read quads 4, 5 against zero weights -- quads 6, 7 are the pad slot and the next pixel)
                    const int kq = (j == CJ - 1 && p.tail8) ? (k & 1) : k;
                    const f16x8 c0 = __builtin_bit_cast(f16x8, xs4[px0 * p.SP + 4 * j + kq]), c1 = __builtin_bit_cast(f16x8, xs4[px1 * p.SP + 4 * j + kq]);
                    const f16x8 aj = __builtin_bit_cast(f16x8, a[j]);
                    // ONE MFMA shape per accumulator chain (an x16 reading what an x32 has just written is a hazard the compiler does not pad): the slot [xh | xl] is the
                    // B operand of both products, A = [wh | wh] gives wh xh + wh xl, A = [wl | 0] gives wl xh (both tuples are per slice, not per tile)
                    const f16x8 ahh = __builtin_shufflevector(aj, aj, 0, 1, 2, 3, 0, 1, 2, 3);
                    const f16x8 al0 = __builtin_shufflevector(aj, f16x8{0, 0, 0, 0, 0, 0, 0, 0}, 4, 5, 6, 7, 8, 9, 10, 11);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahh, c0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahh, c1, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, c0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, c1, acc1, 0, 0, 0);
                    continue;
                }
                if (j == CJ - 1 && p.tail8) { // (wave-uniform) the last 8 channels: k lane kk multiplies channel 16 j + kk, then 16 j + 4 + kk
                    const float* const q0 = xsf + (px0 * p.SP + 4 * j) * 4 + k;
                    const float* const q1 = xsf + (px1 * p.SP + 4 * j) * 4 + k;
                    const float b00 = q0[0], b01 = q0[4], b10 = q1[0], b11 = q1[4];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b00, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b10, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b01, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b11, acc1, 0, 0, 0);
                } else {
                    const float4 b0 = xs4[px0 * p.SP + 4 * j + k], b1 = xs4[px1 * p.SP + 4 * j + k];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b0.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b0.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b0.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b0.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b1.w, acc1, 0, 0, 0);
                }
            }
            [[maybe_unused]] const v2f sc01 = {sc1.x, sc1.y}, sc23 = {sc1.z, sc1.w}, sh01 = {sh1.x, sh1.y}, sh23 = {sh1.z, sh1.w};
            {
#if SNNHIP_IRB_FOLD_BN
                v2f u01 = {acc0[0], acc0[1]}, u23 = {acc0[2], acc0[3]};
                if (S16 && scaled) { u01 *= sxInv; u23 *= sxInv; }
#else
                const v2f u01 = __builtin_elementwise_fma(sc01, v2f{acc0[0], acc0[1]}, sh01), u23 = __builtin_elementwise_fma(sc23, v2f{acc0[2], acc0[3]}, sh23);
#endif
                if constexpr (S16) // (the slice's channels are kept times their weight rows' powers of two: sc1 holds 6 x that power, the depthwise taps its inverse)
                    hs4[k * p.hPlane4 + e0] = make_float4(irb_relu_to(u01[0], sc1.x), irb_relu_to(u01[1], sc1.y), irb_relu_to(u23[0], sc1.z), irb_relu_to(u23[1], sc1.w));
                else
                    hs4[k * p.hPlane4 + e0] = make_float4(irb_act<R6>(p.ac1, u01[0]), irb_act<R6>(p.ac1, u01[1]), irb_act<R6>(p.ac1, u23[0]), irb_act<R6>(p.ac1, u23[1]));
            }
            if (two) {
#if SNNHIP_IRB_FOLD_BN
                v2f u01 = {acc1[0], acc1[1]}, u23 = {acc1[2], acc1[3]};
                if (S16 && scaled) { u01 *= sxInv; u23 *= sxInv; }
#else
                const v2f u01 = __builtin_elementwise_fma(sc01, v2f{acc1[0], acc1[1]}, sh01), u23 = __builtin_elementwise_fma(sc23, v2f{acc1[2], acc1[3]}, sh23);
#endif
                if constexpr (S16)
                    hs4[k * p.hPlane4 + e1] = make_float4(irb_relu_to(u01[0], sc1.x), irb_relu_to(u01[1], sc1.y), irb_relu_to(u23[0], sc1.z), irb_relu_to(u23[1], sc1.w));
                else
                    hs4[k * p.hPlane4 + e1] = make_float4(irb_act<R6>(p.ac1, u01[0]), irb_act<R6>(p.ac1, u01[1]), irb_act<R6>(p.ac1, u23[0]), irb_act<R6>(p.ac1, u23[1]));
            }
        }
        if (more) web += p.wePieces * 64;
#pragma unroll
        for (int j = 0; j < CJ; ++j) a[j] = web[j * 64 + lane]; // the next slice's expand weights, under D / P
        IRBB_ADD(1); // E
        __syncthreads(); // slice c of the hidden tensor is complete; every wave has left D / P (c - 1), whose buffer E (c + 1) will overwrite
        IRBB_ADD(2); // barrier
        // ---- D + P: this wave's output pixel tiles
        const float4* const hb = hs4 + k * p.hPlane4;
#pragma unroll
        for (int g = 0; g < GW; ++g) {
#ifdef SNNHIP_IRB_ABL_NOCONF
            const int hp0 = ((wave + p.NW * g) & 3) * 16 + n16;
#else
            const int hp0 = hpD[g];
#endif
#if SNNHIP_IRB_FOLD_BN
            v2f s01 = {sh2.x, sh2.y}, s23 = {sh2.z, sh2.w};
#else
            v2f s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#endif
#pragma unroll
            for (int fy = 0; fy < 3; ++fy)
#pragma unroll
                for (int fx = 0; fx < 3; ++fx) {
                    const float4 h = hb[hp0 + fy * p.HWd + fx];
                    const float4 w = wd[fy * 3 + fx];
                    s01 = __builtin_elementwise_fma(v2f{h.x, h.y}, v2f{w.x, w.y}, s01);
                    s23 = __builtin_elementwise_fma(v2f{h.z, h.w}, v2f{w.z, w.w}, s23);
                }
#if SNNHIP_IRB_FOLD_BN
            const v2f t01 = s01, t23 = s23;
#else
            const v2f t01 = __builtin_elementwise_fma(v2f{sc2.x, sc2.y}, s01, v2f{sh2.x, sh2.y}), t23 = __builtin_elementwise_fma(v2f{sc2.z, sc2.w}, s23, v2f{sh2.z, sh2.w});
#endif
            const float d0 = irb_act<R6>(p.ac2, t01[0]), d1 = irb_act<R6>(p.ac2, t01[1]), d2 = irb_act<R6>(p.ac2, t23[0]), d3 = irb_act<R6>(p.ac2, t23[1]);
            if constexpr (S16) {
                // (ONE MFMA shape on the output accumulators, irb_image_kernel's note: A = the weight slot [wh | wl], B = [dh | dh], then B = [dl | 0])
                const f16x8 db = split_f16x3(d0, d1, d2, d3);
                const f16x8 dhh = __builtin_shufflevector(db, db, 0, 1, 2, 3, 0, 1, 2, 3);
                const f16x8 dl0 = __builtin_shufflevector(db, f16x8{0, 0, 0, 0, 0, 0, 0, 0}, 4, 5, 6, 7, 8, 9, 10, 11);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ap[cb]), dhh, acc[cb][g], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ap[cb]), dl0, acc[cb][g], 0, 0, 0);
                continue;
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[cb].x, d0, acc[cb][g], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[cb].y, d1, acc[cb][g], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[cb].z, d2, acc[cb][g], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[cb].w, d3, acc[cb][g], 0, 0, 0);
        }
        if (more) wpb += p.wpPieces * 64;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) ap[cb] = wpb[cb * 64 + lane];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) wd[tp] = wpb[NCB * 64 + tp * 4 + k];
        IRBB_ADD(3); // D + P
    }

    // ---- epilogue: lane holds output channels 16 cb + 4 k .. + 3 of its pixels
    float* const yi = y + static_cast<size_t>(img) * p.OH * p.OW * p.Co;
#pragma unroll
    for (int g = 0; g < GW; ++g) {
        const int o = (wave + p.NW * g) * 16 + n16;
        const int oo = tabO[o];
        if (oo < 0) continue;
        const int xr = tabR[o];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int co = cb * 16 + 4 * k;
            if (co >= p.Co) continue;
            const float4 sc = epi3[2 * (cb * 4 + k)], sh = epi3[2 * (cb * 4 + k) + 1];
            float4 r;
            r.x = apply_act<true>(p.ac3, fmaf(sc.x, acc[cb][g][0], sh.x), 0.f);
            r.y = apply_act<true>(p.ac3, fmaf(sc.y, acc[cb][g][1], sh.y), 0.f);
            r.z = apply_act<true>(p.ac3, fmaf(sc.z, acc[cb][g][2], sh.z), 0.f);
            r.w = apply_act<true>(p.ac3, fmaf(sc.w, acc[cb][g][3], sh.w), 0.f);
            if (p.hasRes) { // (the split form's tile is no longer the fp32 input; stride 1 and C == Co: the input pixel sits at the output pixel's offset, in L2)
                const float4 xv = S16 ? *reinterpret_cast<const float4*>(x + static_cast<size_t>(img) * p.H * p.W * p.C + oo + co) : xs4[xr * p.SP + cb * 4 + k];
                r.x = apply_act<true>(p.ac4, r.x + xv.x, 0.f);
                r.y = apply_act<true>(p.ac4, r.y + xv.y, 0.f);
                r.z = apply_act<true>(p.ac4, r.z + xv.z, 0.f);
                r.w = apply_act<true>(p.ac4, r.w + xv.w, 0.f);
            }
            *reinterpret_cast<float4*>(yi + oo + co) = r;
        }
    }
#ifdef SNNHIP_IRBB_TRACE
    IRBB_ADD(4);
    if (btr)
        printf("irbb NCB%d CJ%d GW%d NW%d slices %d MT %d wave %d: setup %llu | E %llu barrier %llu D+P %llu (sums over the slices) | epilogue %llu | total %llu\n", NCB, CJ, GW, p.NW, p.nSlices, MT, wave,
               bAcc[0], bAcc[1], bAcc[2], bAcc[3], bAcc[4], bAcc[0] + bAcc[1] + bAcc[2] + bAcc[3] + bAcc[4]);
#endif
#undef IRBB_ADD
}

struct IrbBandPlan : snnhip_plan {
    IrbBandParams p;
    float* d_we = nullptr;
    float* d_wp = nullptr;
    float* d_e3 = nullptr;
    size_t ldsBytes = 0;
    void (*kernel)(IrbBandParams, const float*, const float4*, const float4*, const float4*, float*) = nullptr;

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "inverted-residual block: expects 1 input (the block input is also the residual), got %d", nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.H && x->w == p.W && x->c == p.C, "irb: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h, x->w, x->c, p.N, p.H, p.W, p.C);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.Co, "irb: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n, out->h, out->w,
                       out->c, p.N, p.OH, p.OW, p.Co);
        SNNHIP_LAUNCH(kernel, dim3(static_cast<unsigned>(p.N * p.nBy * p.nBx)), dim3(static_cast<unsigned>(64 * p.NW)), ldsBytes, ctx->stream, p, x->data,
                      reinterpret_cast<const float4*>(d_we), reinterpret_cast<const float4*>(d_wp), reinterpret_cast<const float4*>(d_e3), out->data);
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

typedef void (*IrbBandFn)(IrbBandParams, const float*, const float4*, const float4*, const float4*, float*);
IrbBandFn pick_irb_band(int ncb, int cj, int gw, bool s16) {
#define SNNHIP_IRBB(NCB_, CJ_) \
    if (ncb == NCB_ && cj == CJ_)  \
        return s16 ? (gw == 1 ? irb_band_kernel<NCB_, CJ_, 1, true, true> : irb_band_kernel<NCB_, CJ_, 2, true, true>) \
                   : (gw == 1 ? irb_band_kernel<NCB_, CJ_, 1, true, false> : irb_band_kernel<NCB_, CJ_, 2, true, false>);
    SNNHIP_IRBB(2, 1)  // 16 -> 96 -> 24 (MobileNetV2 b01)
    SNNHIP_IRBB(2, 2)  // 24 -> 144 -> 24 / 32 (b02, b03), 32 -> 192 -> 32 (b04, b05)
    SNNHIP_IRBB(4, 2)  // 32 -> 192 -> 64 (b06)
#undef SNNHIP_IRBB
    return nullptr;
}

// Band geometry: output rows R x columns SW per block and waves NW, by a small cost model -- MFMA issue slots of the busiest SIMD per round of blocks
// (ceil(tiles / NW) per wave and slice, ceil(NW x blocks per CU / 4) waves on a SIMD) plus a fixed cost per block and per slice -- over the candidates
// that fit 160 KB of LDS.  Returns false if none does.
struct BandChoice {
    int R = 0, SW = 0, NW = 0, GW = 0, MTmax = 0, hPlane4 = 0, wavesPerCU = 0;
    size_t lds = 0;
};
bool choose_band(int N, int H, int W, int C, int Ch, int Co, int OH, int OW, int s, int cus, BandChoice* out) {
    const int SP = C / 4 + 1, nSl = up_div(Ch, 16), NCB = up_div(Co, 16);
    const int eSteps = (C / 16) * 4 + ((C % 16) ? 2 : 0);
    double best = 0.0;
    int pinR = 0, pinSW = 0, pinNW = 0; // SNNHIP_IRB_BAND_GEOM=R,SW,NW pins the geometry (tuning runs, tests of odd shapes)
    if (const char* ge = snnhip::option("SNNHIP_IRB_BAND_GEOM")) sscanf(ge, "%d,%d,%d", &pinR, &pinSW, &pinNW);
    // Measured geometries (round 6, tools/r6_geom.sh: 62 geometries of MobileNetV2's b02 / b03 / b04 / b06 at batch 256 on the round-6 kernel) where the model
    // below ranks them wrong -- it prices eight waves above seven at equal tiles per wave, the kernel runs them 7 % faster: b02 8 x 28 x 7 waves 283 us,
    // x 8 waves 264; b03 2 x 28 x 4 186 us, 4 x 28 x 8 174.  Applied only to those shapes with enough blocks for four rounds.
    if (!pinR && !pinSW && !pinNW && H == 56 && W == 56 && C == 24 && Ch == 144) {
        if (s == 1 && OH == 56 && OW == 56 && N * 14L >= 4L * cus) pinR = 8, pinSW = 28, pinNW = 8;
        if (s == 2 && OH == 28 && OW == 28 && N * 7L >= 4L * cus) pinR = 4, pinSW = 28, pinNW = 8;
    }
    // (fourth session, the split-precision kernel: b04 / b05 7 x 28 x 7 waves 82.2 us against the model's 7 x 28 x 8 at 86.1; b02 / b03 / b06 stay where they are)
    if (!pinR && !pinSW && !pinNW && H == 28 && W == 28 && C == 32 && Ch == 192 && s == 1 && N * 4L >= 4L * cus) pinR = 7, pinSW = 28, pinNW = 7;
    for (int div = 1; div <= 4; div *= 2) {
        const int SW = pinSW ? std::min(pinSW, OW) : up_div(OW, div);
        if (div > 1 && (SW < 14 || pinSW)) break;
        for (int R = 1; R <= std::min(OH, 16); ++R) {
            if (pinR && R != std::min(pinR, OH)) continue;
            const int HH = (R - 1) * s + 3, HWd = (SW - 1) * s + 3;
            const int xPx = std::min(HH, H) * std::min(HWd, W), MT = up_div(xPx, 16), OT = up_div(R * SW, 16);
            const int hPlane4 = round_up(HH * HWd + 1, 16);
            for (int NW = 4; NW <= 8; ++NW) {
                if (pinNW ? NW != pinNW : (NW == 5 || NW == 6)) continue;
                if (!pinNW && !pinR && s == 2 && (NW != 4 || R > 4)) continue; // stride 2 (measured, b03 / b06): four waves and at most four rows, e.g. b03 181 us vs 196-229 with 5-7 rows
                const int GW = up_div(OT, NW);
                if (GW > 2) continue;
                const size_t lds = (static_cast<size_t>(std::max(MT * 16 * SP, round_up(xPx * SP, 64))) + 8 * hPlane4) * 16 + (static_cast<size_t>(MT) * 16 + 3 * NW * GW * 16) * 4;
                if (lds > 160 * 1024) continue;
                const int bpc = std::max(1, std::min(std::min(4, 16 / NW), static_cast<int>((160 * 1024) / lds)));
                const long blocks = static_cast<long>(N) * up_div(OH, R) * up_div(OW, SW);
                // Fitted to 31 measured geometries of MobileNetV2 b01-b06 at batch 256 (tools/sweep_band.sh; within ~15 %): a wave's matrix work and its VALU
                // work add (they share the issue port), a SIMD runs its ceil(waves / 4) waves one after the other and pays ~150 cycles per slice and
                // wave on the CU (barrier, LDS and issue contention), and a round of blocks cannot be shorter than one wave's work plus its exposed
                // latencies (staging 8 000 cycles, 3 000 per slice).  Only the measured wave counts (4, 7, 8) are candidates: 6 waves ran 25 % slower than predicted.
                const double work = nSl * (32.0 * (up_div(MT, NW) * eSteps + GW * NCB * 4) + 100.0 * up_div(MT, NW) + 150.0 * GW);
                const double lat = 8000.0 + nSl * 3000.0;
                const double rounds = static_cast<double>(blocks) / (static_cast<double>(cus) * bpc);
                const double cost = rounds * std::max(work * up_div(NW * bpc, 4) + nSl * 150.0 * NW * bpc, work + lat);
                if (best == 0.0 || cost < best) {
                    best = cost;
                    *out = BandChoice{R, SW, NW, GW, MT, hPlane4, NW * bpc, lds};
                }
            }
        }
    }
    return best > 0.0;
}

typedef void (*IrbFn)(IrbParams, const float*, const float4*, const float4*, const float4*, float*);
template <int G, bool R6, bool S16 = false>
IrbFn pick_irb_wave(int ncb, int cj) {
#define SNNHIP_IRBW(NCBT_, CJT_) \
    if (ncb <= NCBT_ && cj <= CJT_) return irb_wave_kernel<G, NCBT_, CJT_, R6, false, S16>;
    SNNHIP_IRBW(2, 1)
    SNNHIP_IRBW(2, 2)
    SNNHIP_IRBW(4, 2)
    SNNHIP_IRBW(4, 4)
    SNNHIP_IRBW(6, 4)
    SNNHIP_IRBW(6, 6)
    if constexpr (G == 1) { // the 14x14 / 7x7 blocks (SNNHIP_IRB_FUSION=all: slower than their separate layers, kept for the parity tests)
        SNNHIP_IRBW(10, 6)
        SNNHIP_IRBW(10, 10)
        SNNHIP_IRBW(20, 10)
    }
#undef SNNHIP_IRBW
    return nullptr;
}
} // namespace

// expand / dw / project: the three per-layer plans (borrowed; only read here; expand may be null: DepthwiseConv2D -> Conv2D 1x1, the expansion-factor-1
// block at the head of MobileNetV2); add: the residual Add plan or nullptr.
// stem (with expand == null, add == null): the 3x3 convolution of a 3-channel image in front of the depthwise layer (MobileNetV2: Conv2D 3x3 s2 3->32 ->
// DepthwiseConv2D -> Conv2D 1x1) takes the expand layer's place: K = 27 image values per output pixel, gathered by the staging (IrbParams::stemK).
int make_irb_plan(snnhip_ctx* ctx, snnhip_plan* expandPlan, snnhip_plan* dwPlan, snnhip_plan* projectPlan, snnhip_plan* addPlan, snnhip_plan** out,
                  snnhip_plan* stemPlan) {
    if (snnhip::option("SNNHIP_NO_IRB_FUSION")) return SNNHIP_E_UNSUPPORTED;
    auto* cs = stemPlan ? dynamic_cast<ConvPlanBase*>(stemPlan) : nullptr;
    if (stemPlan) {
        // Opt-in (SNNHIP_STEM_IRB_FUSION=1): MobileNetV2's head at batch 256 takes 403 us fused, 157 + 237 us as stem + two-layer kernel -- the 0.82 GB
        // of traffic it removes buy no time, the kernel is bound by instruction issue and the halo tile makes the stem do 1.9x its work
        const char* on = snnhip::option("SNNHIP_STEM_IRB_FUSION");
        if (!cs || expandPlan || addPlan || !on || atoi(on) == 0) return SNNHIP_E_UNSUPPORTED;
        const ConvGeom& gs = cs->g;
        if (cs->depthwise || gs.kh != 3 || gs.kw != 3 || gs.IC != 3 || gs.sh != gs.sw || gs.sh < 1 || gs.sh > 2 || gs.preMode != 0 || gs.addAct >= 0 ||
            gs.dtype != SNNHIP_F32 || (gs.padMode != SNNHIP_PAD_CONSTANT && gs.padMode != SNNHIP_PAD_NONE) || gs.OC % 16 != 0)
            return SNNHIP_E_UNSUPPORTED;
    }
    const char* irbMode = snnhip::option("SNNHIP_IRB_FUSION"); // "all": also the 14x14 / 7x7 blocks, where the separate layers are faster (tools/bench_irb.py)
    auto* ce = dynamic_cast<ConvPlanBase*>(expandPlan);
    auto* cd = dynamic_cast<ConvPlanBase*>(dwPlan);
    auto* cp = dynamic_cast<ConvPlanBase*>(projectPlan);
    auto* ad = addPlan ? dynamic_cast<EltwisePlanBase*>(addPlan) : nullptr;
    const bool noExpand = expandPlan == nullptr && !cs; // DepthwiseConv2D -> Conv2D 1x1 (MobileNetV2's first block, expansion factor 1)
    if ((!noExpand && !ce && !cs) || !cd || !cp || (ce && ce->depthwise) || !cd->depthwise || cp->depthwise || (addPlan && (!ad || ad->mode != 0))) return SNNHIP_E_UNSUPPORTED;
    ConvGeom geId = cd->g; // stand-in geometry of the missing expand layer: identity on the depthwise layer's input
    geId.kh = geId.kw = geId.sh = geId.sw = 1;
    geId.OC = geId.IC;
    geId.OH = geId.H;
    geId.OW = geId.W;
    geId.act = SNNHIP_ACT_NONE;
    geId.useBN = 0;
    geId.preMode = 0;
    geId.addAct = -1;
    if (cs) { // the stem as a pointwise layer over its im2col'd input: 27 (+1 zero) 'channels' -> OC, on the depthwise layer's grid
        geId.IC = 28;
        geId.OC = cs->g.OC;
        geId.act = cs->g.act;
        geId.leaky = cs->g.leaky;
        geId.useBN = cs->g.useBN;
        if (cs->g.N != cd->g.N || cs->g.OH != cd->g.H || cs->g.OW != cd->g.W) return SNNHIP_E_UNSUPPORTED;
    }
    const ConvGeom &ge = (noExpand || cs) ? geId : ce->g, &gd = cd->g, &gp = cp->g;
    if (noExpand && gd.IC % 16 != 0) return SNNHIP_E_UNSUPPORTED; // whole 16-channel slices of the x tile
    auto pointwise = [](const ConvGeom& g) { return g.kh == 1 && g.kw == 1 && g.sh == 1 && g.sw == 1 && g.preMode == 0 && g.addAct < 0 && g.dtype == SNNHIP_F32; };
    if (!pointwise(ge) || !pointwise(gp) || gd.dtype != SNNHIP_F32 || gd.kh != 3 || gd.kw != 3 || gd.sh != gd.sw || gd.sh < 1 || gd.sh > 2 || gd.preMode != 0)
        return SNNHIP_E_UNSUPPORTED;
    if (gd.padMode != SNNHIP_PAD_CONSTANT && gd.padMode != SNNHIP_PAD_NONE) return SNNHIP_E_UNSUPPORTED;
    if (ge.OC != gd.IC || gd.OC != gp.IC || ge.N != gd.N || gd.N != gp.N || ge.OH != gd.H || ge.OW != gd.W || gd.OH != gp.H || gd.OW != gp.W || ge.OH != ge.H ||
        ge.OW != ge.W || gp.OH != gp.H || gp.OW != gp.W)
        return SNNHIP_E_UNSUPPORTED;
    if (gd.padx < 0 || gd.padx > 2 || gd.pady < 0 || gd.pady > 2) return SNNHIP_E_UNSUPPORTED;
    const int C = ge.IC, Ch = ge.OC, Co = gp.OC, s = gd.sh;
    // Where it pays: the blocks whose expanded tensor is big (MobileNetV2 b01-b06, 112x112 .. 28x28 inputs).  Batch 256, us, separate layers /
    // this kernel: b01 (112x112, stride 2) 787 / 364, b02 (56x56) 615 / 334, b03 (56x56 s2) 368 / 237, b04 (28x28) 201 / 129, b06 (28x28 s2) 124 / 99;
    // from 14x14 down the separate layers win (b07 143 / 148, b10 149 / 170, b11 258 / 332): few tiles per image and 24-60 slices of weights.
    const bool fuseAll = irbMode && strcmp(irbMode, "all") == 0;
    if (C % 4 || Ch % 4 || Co % 4 || C > 16 * kMaxCj || Co > 16 * kMaxNCB) return SNNHIP_E_UNSUPPORTED;
    const int acts[4] = {ge.act, gd.act, gp.act, ad ? ad->d.act : 0};
    for (int a : acts)
        if (!act_is_simple(a)) return SNNHIP_E_UNSUPPORTED;
    if (ad && (s != 1 || C != Co || ad->d.N != gp.N || ad->d.H != gp.OH || ad->d.W != gp.OW || ad->d.C != Co)) return SNNHIP_E_UNSUPPORTED;
    if (static_cast<double>(ge.N) * ge.H * ge.W * std::max(C, Ch) >= 2147483647.0) return SNNHIP_E_UNSUPPORTED;
    // The whole-image kernel (irb_image_kernel): maps of at most 13 pixel tiles (14x14), whole 16-channel blocks in and out, hidden channels in four equal
    // quarters of whole slices, ReLU6 after expand and depthwise, and at least one image per two CUs (a block is an image).  SNNHIP_IRB_IMAGE=0 switches it
    // off, =1 takes it at any batch size (tests).
    IrbImgFn imgFn = nullptr;
    int imgOS = 1;
    // SNNHIP_IRB_SPLIT=0: the whole-image kernel's pointwise stages as fp32 MFMAs (round 5's form); default: the split-precision form (three f16 products per fp32 product)
    const char* splitOpt = snnhip::option("SNNHIP_IRB_SPLIT");
    const bool imgS16 = !(splitOpt && atoi(splitOpt) == 0) && !SNNHIP_IRBI_FOLD_BN;
    const bool bandS16 = !(splitOpt && atoi(splitOpt) == 0) && SNNHIP_IRB_FOLD_BN; // (band / wave kernels: the hidden slice stays scaled per channel, which needs the folded epilogue)
    IrbImgParams ip = {};
    size_t imgLds = 0;
    std::vector<int> imgTabs;
    {
        const char* io = snnhip::option("SNNHIP_IRB_IMAGE");
        const int mode = io ? atoi(io) : -1;
        const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
        const int HW = ge.H * ge.W, OHW = gd.OH * gd.OW;
        int GT = 0;
        if (mode != 0 && !cs && !noExpand && !fuseAll && C % 16 == 0 && Co % 16 == 0 && Ch % 64 == 0 && HW <= 16 * 13 && ge.act == SNNHIP_ACT_RELU6 &&
            gd.act == SNNHIP_ACT_RELU6 && (ge.N * 2 >= cus || mode == 1))
        {
            imgFn = pick_irb_image(Co / 16, C / 16, up_div(OHW, 16), imgS16, &GT);
            // no instantiation with that many output blocks (MobileNetV2's b16: 20): two blocks per image, ten output blocks each, each with its own expand / depthwise
            // pass (1.34x the block's flops) -- separate layers 172 us, this way 2 x b14's 62 (SNNHIP_IRB_IMAGE_HALVES=0 keeps the separate layers)
            const char* hv = snnhip::option("SNNHIP_IRB_IMAGE_HALVES");
            if (!imgFn && Co % 32 == 0 && !ad && !(hv && atoi(hv) == 0)) {
                imgFn = pick_irb_image(Co / 32, C / 16, up_div(OHW, 16), imgS16, &GT);
                if (imgFn) imgOS = 2;
            }
        }
        if (imgFn) {
            ip.N = ge.N; ip.H = ge.H; ip.W = ge.W; ip.C = C; ip.Ch = Ch; ip.Co = Co; ip.OH = gd.OH; ip.OW = gd.OW; ip.s = s;
            ip.HW = HW;
            ip.OHW = OHW;
            ip.MT = up_div(HW, 16);
            ip.SP = C / 4 + 1;
            ip.magicSP = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(ip.SP) - 1) / static_cast<unsigned>(ip.SP));
            const int rows = std::max(gd.pady + ge.H, (gd.OH - 1) * s + 3);
            ip.HWd = std::max(gd.padx + ge.W, (gd.OW - 1) * s + 3);
            ip.hPlane4 = round_up(std::max(rows * ip.HWd + 1, GT * 16), 16); // (... and room for the reduction: 4 waves x GT tiles x 64 float4 in 16 planes)
            ip.offH4 = std::max(ip.MT * 16 * ip.SP, round_up(HW * ip.SP, 64));
            ip.slicesPerWave = Ch / 64;
            ip.hasRes = ad ? 1 : 0;
            ip.OS = imgOS;
            imgLds = (static_cast<size_t>(ip.offH4) + 16 * ip.hPlane4) * 16 + static_cast<size_t>(ip.MT + GT) * 16 * 4 + (imgS16 ? 16 : 0); // (+ the split form's magnitude slot)
            if (imgLds > 160 * 1024 || HW * ip.SP >= 65536 ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(imgFn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(imgLds)) != hipSuccess)
                imgFn = nullptr;
            ip.wePieces = C / 16 + 1;
            ip.wpPieces = Co / 16 + 1;
            if (imgFn) {
                ip.ac1 = make_act_cfg(ge.act, ge.leaky);
                ip.ac2 = make_act_cfg(gd.act, gd.leaky);
                ip.ac3 = make_act_cfg(gp.act, gp.leaky);
                ip.ac4 = make_act_cfg(ad ? ad->d.act : 0, ad ? ad->d.leaky : 0.0f);
            }
            // pixel tables: hidden position of x-tile pixel i (padding pixels: the plane's last slot), of output pixel o's first tap (padding pixels: 0)
            if (imgFn) {
                std::vector<int>& tb = imgTabs;
                tb.assign(static_cast<size_t>(ip.MT + GT) * 16, 0);
                for (int i = 0; i < ip.MT * 16; ++i) tb[i] = i < HW ? (i / ge.W + gd.pady) * ip.HWd + i % ge.W + gd.padx : ip.hPlane4 - 1;
                for (int o = 0; o < GT * 16; ++o) tb[ip.MT * 16 + o] = o < OHW ? (o / gd.OW) * s * ip.HWd + (o % gd.OW) * s : 0;
            }
        }
    }
    if (!imgFn && !fuseAll && ge.H * ge.W < 28 * 28) return SNNHIP_E_UNSUPPORTED;
    // The band kernel (irb_band_kernel) on the larger maps: 16 k or 16 k + 8 input channels, at most 64 output channels, ReLU6 after expand and depthwise,
    // at least two blocks per CU, and nine or more slices: measured at batch 256 (tools/sweep_band.sh, us, irb_wave_kernel / this kernel at its best geometry):
    // b02 311 / 301, b03 218 / 182, b04 126 / 103, b06 96 / 79; b01 (six slices, 16 channels in) 351 / 446 -- its blocks are too short for their fixed cost.
    // SNNHIP_IRB_BAND=0 keeps irb_wave_kernel, =1 takes the band kernel wherever it runs (tests).
    IrbBandFn bandFn = nullptr;
    BandChoice bc;
    if (!imgFn && !cs && !noExpand && !fuseAll) {
        const char* bo = snnhip::option("SNNHIP_IRB_BAND");
        const int mode = bo ? atoi(bo) : -1;
        const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
        if (mode != 0 && (C % 16 == 0 || C % 16 == 8) && ge.act == SNNHIP_ACT_RELU6 && gd.act == SNNHIP_ACT_RELU6 &&
            choose_band(ge.N, ge.H, ge.W, C, Ch, Co, gd.OH, gd.OW, s, cus, &bc) &&
            ((static_cast<long>(ge.N) * up_div(gd.OH, bc.R) * up_div(gd.OW, bc.SW) >= 2L * cus && Ch >= 144) || mode == 1) &&
            bc.MTmax * 16 * (C / 4 + 1) < 65536)
            bandFn = pick_irb_band(up_div(Co, 16), up_div(C, 16), bc.GW, bandS16);
        if (bandFn && bandS16) bc.lds += 16; // (the split form's magnitude slot behind the tables)
        if (bandFn && hipFuncSetAttribute(reinterpret_cast<const void*>(bandFn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bc.lds)) != hipSuccess) bandFn = nullptr;
    }
    const bool tail8 = !imgFn && !cs && !noExpand && C % 16 == 8 && !snnhip::option("SNNHIP_IRB_NO_TAIL8"); // (the switch: A/B runs of irb_wave_kernel)
    const bool r6 = ge.act == SNNHIP_ACT_RELU6 && gd.act == SNNHIP_ACT_RELU6;
    const bool waveS16 = !imgFn && !bandFn && !cs && !noExpand && r6 && bandS16; // (irb_wave_kernel: MobileNetV2's b01)
    const bool s16 = (imgFn && imgS16) || (bandFn && bandS16) || waveS16;

    IrbParams p = {};
    p.N = ge.N; p.H = ge.H; p.W = ge.W; p.C = C; p.Ch = Ch; p.Co = Co; p.OH = gd.OH; p.OW = gd.OW; p.s = s; p.padx = gd.padx; p.pady = gd.pady;
    p.Cj = up_div(C, 16);
    p.magicQuads = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(4 * p.Cj) - 1) / static_cast<unsigned>(4 * p.Cj));
    p.NCB = up_div(Co, 16);
    // tile per wave: 2 pixel groups (4x8) for stride 1, 1 (2x8) for stride 2 -- measured (tools/gpu_irb.sh): the small tiles' occupancy beats their extra
    // halo work; a smaller tile when the per-wave LDS would leave fewer than 4 waves on a CU.  SNNHIP_IRB_WAVE_G=1|2|4 pins it (tests).
    const char* gopt = snnhip::option("SNNHIP_IRB_WAVE_G");
    int G = gopt ? atoi(gopt) : (s == 2 ? 1 : 2);
    if (G != 1 && G != 2 && G != 4) G = s == 2 ? 1 : 2;
    int NWv = 0, perWave = 0;
    for (; !imgFn && !bandFn; G >>= 1) {
        const int TH = 2 * G, TWv = 8;
        p.TWs = 3;
        p.HH = (TH - 1) * s + 3;
        p.HWd = (TWv - 1) * s + 3;
        p.magicHWd = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(p.HWd) - 1) / static_cast<unsigned>(p.HWd));
        p.HP = p.HH * p.HWd;
        p.MT = up_div(p.HP, 16);
        p.tilesX = up_div(p.OW, TWv);
        p.tilesY = up_div(p.OH, TH);
        p.xPlane = round_up(p.MT * 16 * 4, 64);
        p.hPlane = p.xPlane;
        p.offH = 4 * p.Cj * p.xPlane;   // the wave's hidden slice (4 quad planes) behind its x planes
        p.offWe = p.offH + (noExpand ? 0 : 4 * p.hPlane); // ... and its inside-the-image mask
        perWave = round_up(p.offWe + p.MT * 16, 64);
        p.offMask = perWave;             // floats per wave
        // waves per block: the block size (2..4 waves) that puts the most waves on a CU's 160 KB of LDS
        int bestWaves = 0;
        for (int nw = 4; nw >= 2; --nw) {
            const int blocks = std::min(8, static_cast<int>((160 * 1024) / (static_cast<size_t>(nw) * perWave * sizeof(float))));
            if (nw * blocks > bestWaves) {
                bestWaves = nw * blocks;
                NWv = nw;
            }
        }
        if (bestWaves >= 4 || G == 1) {
            if (bestWaves < 2) return SNNHIP_E_UNSUPPORTED; // the x tile of one wave does not fit LDS
            break;
        }
    }
    p.nChunks = up_div(Ch, 16);
    p.wePieces = p.Cj + 1;
    p.wpPieces = p.NCB + 1;
    p.offWp = 0;
    p.hasRes = ad ? 1 : 0;
    p.tail8 = tail8 ? 1 : 0;
    p.noExpand = noExpand ? 1 : 0;
    if (cs) {
        p.stemK = 27;
        p.stemS = cs->g.sh;
        p.stemPadX = cs->g.padx;
        p.stemPadY = cs->g.pady;
        p.IH = cs->g.H;
        p.IW = cs->g.W;
        if (static_cast<double>(cs->g.N) * cs->g.H * cs->g.W * 3 >= 2147483647.0) return SNNHIP_E_UNSUPPORTED;
        p.rawH = (p.HH - 1) * p.stemS + 3;
        p.rawW3 = ((p.HWd - 1) * p.stemS + 3) * 3;
        p.magicRawW3 = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(p.rawW3) - 1) / static_cast<unsigned>(p.rawW3));
        if (p.rawH * p.rawW3 > 4 * p.hPlane || p.rawW3 < 33 || p.HWd < 8) return SNNHIP_E_UNSUPPORTED; // the patch is parked in the hidden-slice region until the x tile is built
    }
    p.ac1 = make_act_cfg(ge.act, ge.leaky);
    p.ac2 = make_act_cfg(gd.act, gd.leaky);
    p.ac3 = make_act_cfg(gp.act, gp.leaky);
    p.ac4 = make_act_cfg(ad ? ad->d.act : 0, ad ? ad->d.leaky : 0.0f);
    const size_t lds = static_cast<size_t>(NWv) * perWave * sizeof(float);
    IrbFn fn = nullptr;
    if (imgFn || bandFn) {
    } else if (cs) { // stem mode: its own instantiations (Cj = 2: the 27 image values; up to 32 output channels)
        if (p.NCB > 2) return SNNHIP_E_UNSUPPORTED;
        if (G == 4) fn = r6 ? irb_wave_kernel<4, 2, 2, true, true> : irb_wave_kernel<4, 2, 2, false, true>;
        if (G == 2) fn = r6 ? irb_wave_kernel<2, 2, 2, true, true> : irb_wave_kernel<2, 2, 2, false, true>;
        if (G == 1) fn = r6 ? irb_wave_kernel<1, 2, 2, true, true> : irb_wave_kernel<1, 2, 2, false, true>;
    } else if (G == 4) fn = waveS16 ? pick_irb_wave<4, true, true>(p.NCB, p.Cj) : r6 ? pick_irb_wave<4, true>(p.NCB, p.Cj) : pick_irb_wave<4, false>(p.NCB, p.Cj);
    else if (G == 2) fn = waveS16 ? pick_irb_wave<2, true, true>(p.NCB, p.Cj) : r6 ? pick_irb_wave<2, true>(p.NCB, p.Cj) : pick_irb_wave<2, false>(p.NCB, p.Cj);
    else if (G == 1) fn = waveS16 ? pick_irb_wave<1, true, true>(p.NCB, p.Cj) : r6 ? pick_irb_wave<1, true>(p.NCB, p.Cj) : pick_irb_wave<1, false>(p.NCB, p.Cj);
    if (!fn && !imgFn && !bandFn) return SNNHIP_E_UNSUPPORTED;
    if (fn && lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
        set_error("irb_fused: hipFuncSetAttribute(%zu) failed", lds);
        return SNNHIP_E_HIP;
    }

    // ---- slice blobs (the kernel's LDS images)
    const std::vector<float> e1 = noExpand ? std::vector<float>(static_cast<size_t>(Ch) * 2, 0.0f) : fold_epilogue(cs ? cs->epi4 : ce->epi4, Ch, ge.useBN);
    const std::vector<float> e2 = fold_epilogue(cd->epi4, Ch, gd.useBN), e3 = fold_epilogue(cp->epi4, Co, gp.useBN);
    std::vector<float> we(static_cast<size_t>(p.nChunks) * p.wePieces * 256, 0.0f), wp(static_cast<size_t>(p.nChunks) * p.wpPieces * 256, 0.0f);
    // split-precision form (irb_image_kernel<.., S16>): every weight row times the power of two that puts its largest magnitude into [2^13, 2^14) -- undone in the
    // row's epilogue scale --, each weight as [hi | lo] halves: a lane's 16-byte operand slot holds {hi of its four K values, lo of the same four}
    std::vector<int> rowExpE(static_cast<size_t>(Ch), 0), rowExpP(static_cast<size_t>(Co), 0);
    if (s16) {
        auto row_exp = [](const float* w, int n, int stride) {
            float m = 0.0f;
            for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(w[static_cast<size_t>(i) * stride]));
            if (!(m > 0.0f) || !std::isfinite(m)) return 0;
            int ex = 0;
            (void)std::frexp(m, &ex); // m = f 2^ex, f in [0.5, 1)
            return std::min(100, std::max(-100, 14 - ex));
        };
        const bool fold = imgFn ? SNNHIP_IRBI_FOLD_BN != 0 : SNNHIP_IRB_FOLD_BN != 0;
        for (int hc = 0; hc < Ch; ++hc) {
            rowExpE[hc] = row_exp(ce->w_oihw.data() + static_cast<size_t>(hc) * C, C, 1);
            if (fold && e1[2 * hc] != 0.0f && std::isfinite(e1[2 * hc])) { // (the row the kernel multiplies is scale x weights)
                int ex = 0;
                (void)std::frexp(std::fabs(e1[2 * hc]), &ex);
                rowExpE[hc] = std::min(100, std::max(-100, rowExpE[hc] - (ex - 1)));
            }
        }
        for (int co = 0; co < Co; ++co) rowExpP[co] = row_exp(cp->w_oihw.data() + static_cast<size_t>(co) * Ch, Ch, 1);
    }
    auto put_split = [](float* slot, int jj, float w) { // slot: 4 floats = 8 halves
        _Float16* const h = reinterpret_cast<_Float16*>(slot);
        const _Float16 hi = static_cast<_Float16>(w);
        h[jj] = hi;
        h[4 + jj] = static_cast<_Float16>(w - static_cast<float>(hi));
    };
    for (int c = 0; c < p.nChunks; ++c) {
        float* wb = we.data() + static_cast<size_t>(c) * p.wePieces * 256;
        float* pb = wp.data() + static_cast<size_t>(c) * p.wpPieces * 256;
        for (int m = 0; m < 16; ++m) {
            const int hc = 16 * c + m;
            if (hc >= Ch) continue;
            // the layers' folded scales go into their weights where the kernel starts its sums from the shift (irb_wave / irb_band; irb_image: SNNHIP_IRBI_FOLD_BN)
            const bool foldBN = imgFn ? SNNHIP_IRBI_FOLD_BN != 0 : SNNHIP_IRB_FOLD_BN != 0;
            const float s1 = (foldBN && !noExpand) ? e1[2 * hc] : 1.0f, s2 = foldBN ? e2[2 * hc] : 1.0f;
            // expand: [j][lane = 16 kk + m] float4 {We[hc][16 j + 4 kk + jj]}
            for (int ic = 0; ic < C && !noExpand; ++ic) {
                const int j = ic / 16, kk = (ic % 16) / 4, jj = ic % 4;
                if (!s16 && tail8 && j == p.Cj - 1) { // irb_band_kernel's last 8 channels: component 0 = channel 16 j + kk, component 1 = channel 16 j + 4 + kk
                    wb[j * 256 + ((ic % 4) * 16 + m) * 4 + (ic % 16) / 4] = s1 * ce->w_oihw[static_cast<size_t>(hc) * C + ic];
                    continue;
                }
                if (cs) { // 'channel' ic = 3 tap + c of the im2col'd image (the staging's order); the stem's weights are [oc][c][tap]
                    if (ic < 27) wb[j * 256 + (kk * 16 + m) * 4 + jj] = s1 * cs->w_oihw[(static_cast<size_t>(hc) * 3 + ic % 3) * 9 + ic / 3];
                    continue;
                }
                if (s16) { // (the band kernel's tail8 component order does not apply: its split form reads whole K groups)
                    put_split(wb + j * 256 + (kk * 16 + m) * 4, jj, std::ldexp(s1 * ce->w_oihw[static_cast<size_t>(hc) * C + ic], rowExpE[hc]));
                    continue;
                }
                wb[j * 256 + (kk * 16 + m) * 4 + jj] = s1 * ce->w_oihw[static_cast<size_t>(hc) * C + ic];
            }
            // split form, unfolded epilogue (image kernel): the row's power of two is undone in the multiplier; folded (band kernel): the slice stays scaled -- the sums
            // start from the scaled shift, ReLU6 clamps to the scaled 6 (kept in the multiplier's slot), the depthwise taps carry the inverse power
            const bool keepScaled = s16 && foldBN;
            wb[p.Cj * 256 + m] = keepScaled ? std::ldexp(6.0f, rowExpE[hc]) : s16 ? std::ldexp(e1[2 * hc], -rowExpE[hc]) : e1[2 * hc];
            wb[p.Cj * 256 + 16 + m] = keepScaled ? std::ldexp(e1[2 * hc + 1], rowExpE[hc]) : e1[2 * hc + 1];
            // depthwise taps [tap][16], scale[16], shift[16]
            for (int tp = 0; tp < 9; ++tp) pb[p.NCB * 256 + tp * 16 + m] = keepScaled ? std::ldexp(s2 * cd->w_oihw[static_cast<size_t>(hc) * 9 + tp], -rowExpE[hc]) : s2 * cd->w_oihw[static_cast<size_t>(hc) * 9 + tp];
            pb[p.NCB * 256 + 144 + m] = e2[2 * hc];
            pb[p.NCB * 256 + 160 + m] = e2[2 * hc + 1];
        }
        // project: [cb][lane = 16 kk + m] float4 {Wp[co = 16 cb + m][hc = 16 c + 4 kk + jj]}
        for (int co = 0; co < Co; ++co)
            for (int q = 0; q < 16; ++q) {
                const int hc = 16 * c + q;
                if (hc >= Ch) continue;
                if (s16) {
                    put_split(pb + (co / 16) * 256 + ((q / 4) * 16 + co % 16) * 4, q % 4, std::ldexp(cp->w_oihw[static_cast<size_t>(co) * Ch + hc], rowExpP[co]));
                    continue;
                }
                pb[(co / 16) * 256 + ((q / 4) * 16 + co % 16) * 4 + q % 4] = cp->w_oihw[static_cast<size_t>(co) * Ch + hc];
            }
    }
    // final epilogue: per (cb, k) {scale float4, shift float4}
    std::vector<float> e3p(static_cast<size_t>(p.NCB) * 4 * 8, 0.0f);
    for (int co = 0; co < Co; ++co) {
        e3p[(co / 4) * 8 + co % 4] = s16 ? std::ldexp(e3[2 * co], -rowExpP[co]) : e3[2 * co];
        e3p[(co / 4) * 8 + 4 + co % 4] = e3[2 * co + 1];
    }

    const double inElems = cs ? static_cast<double>(p.N) * p.IH * p.IW * 3 : static_cast<double>(p.N) * p.H * p.W * C;
    const double fusedBytes = 4.0 * (inElems + static_cast<double>(p.N) * p.OH * p.OW * Co + static_cast<double>(Ch) * (C + Co + 9));
    if (bandFn) {
        auto* bpl = new IrbBandPlan();
        IrbBandParams& b = bpl->p;
        b = IrbBandParams{};
        b.N = ge.N; b.H = ge.H; b.W = ge.W; b.C = C; b.Ch = Ch; b.Co = Co; b.OH = gd.OH; b.OW = gd.OW; b.s = s; b.padx = gd.padx; b.pady = gd.pady;
        b.R = bc.R; b.SW = bc.SW; b.nBy = up_div(gd.OH, bc.R); b.nBx = up_div(gd.OW, bc.SW);
        b.HWd = (bc.SW - 1) * s + 3;
        b.hPlane4 = bc.hPlane4;
        b.SP = C / 4 + 1;
        b.magicSP = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(b.SP) - 1) / static_cast<unsigned>(b.SP));
        b.MTmax = bc.MTmax;
        b.offH4 = std::max(bc.MTmax * 16 * b.SP, round_up(std::min((bc.R - 1) * s + 3, ge.H) * std::min(b.HWd, ge.W) * b.SP, 64));
        b.offTab4 = b.offH4 + 8 * b.hPlane4;
        b.NW = bc.NW;
        b.nSlices = p.nChunks; b.wePieces = p.wePieces; b.wpPieces = p.wpPieces;
        b.hasRes = ad ? 1 : 0;
        b.tail8 = tail8 ? 1 : 0;
        b.ac1 = make_act_cfg(ge.act, ge.leaky);
        b.ac2 = make_act_cfg(gd.act, gd.leaky);
        b.ac3 = make_act_cfg(gp.act, gp.leaky);
        b.ac4 = make_act_cfg(ad ? ad->d.act : 0, ad ? ad->d.leaky : 0.0f);
        bpl->ctx = ctx;
        bpl->kernel = bandFn;
        bpl->ldsBytes = bc.lds;
        if (const char* padOpt = snnhip::option("SNNHIP_IRB_BAND_LDS_PAD")) { // developer switch: extra dynamic LDS per block (residency experiments)
            bpl->ldsBytes += static_cast<size_t>(atoi(padOpt));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bandFn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bpl->ldsBytes));
        }
        bpl->dtype = SNNHIP_F32;
        int rc = bpl->upload(we.data(), we.size(), &bpl->d_we);
        if (rc == SNNHIP_OK) rc = bpl->upload(wp.data(), wp.size(), &bpl->d_wp);
        if (rc == SNNHIP_OK) rc = bpl->upload(e3p.data(), e3p.size(), &bpl->d_e3);
        if (rc != SNNHIP_OK) {
            delete bpl;
            return rc;
        }
        memcpy(bpl->inDims, expandPlan->inDims, sizeof(bpl->inDims));
        memcpy(bpl->outDims, projectPlan->outDims, sizeof(bpl->outDims));
        bpl->flops = ce->flops + cd->flops + cp->flops;
        bpl->bytes = ce->bytes + cd->bytes + cp->bytes + (addPlan ? addPlan->bytes : 0.0); // unfused accounting of the layers it replaces
        bpl->kernelBytes = fusedBytes;
        char bb[352];
        snprintf(bb, sizeof(bb), "irb_fused_mfma_%s [conv1x1 %d->%d + depthwise3x3 s%d + conv1x1 %d->%d%s] band per block (%d rows x %d cols, %d waves, %d px tiles in), "
                 "slices=%d lds=%zuB hbm_bytes=%.6g kernel=irb_band_kernel<%d,%d,%d,true,%s> relu6-epilogues%s",
                 bandS16 ? "f16x3split_16x16x32" : "f32_16x16x4", C, Ch, s, Ch, Co, addPlan ? " + add" : "", bc.R, bc.SW, bc.NW, bc.MTmax, p.nChunks, bc.lds, fusedBytes, up_div(Co, 16),
                 up_div(C, 16), bc.GW, bandS16 ? "true" : "false", tail8 ? " tail8" : "");
        bpl->desc = bb;
        if (snnhip::option("SNNHIP_IRB_OCC")) { // developer switch: what the runtime says about co-resident blocks of this geometry
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(bandFn), 64 * bc.NW, bc.lds);
            fprintf(stderr, "[irb occ] band R=%d SW=%d NW=%d GW=%d lds=%zu blocks=%ld: %d blocks per CU (model: %d waves)\n", bc.R, bc.SW, bc.NW, bc.GW, bc.lds,
                    static_cast<long>(b.N) * b.nBy * b.nBx, nb, bc.wavesPerCU);
        }
        *out = bpl;
        return SNNHIP_OK;
    }
    if (imgFn) {
        auto* ipl = new IrbImagePlan();
        ipl->ctx = ctx;
        ipl->p = ip;
        ipl->kernel = imgFn;
        ipl->ldsBytes = imgLds;
        ipl->dtype = SNNHIP_F32;
        int rc = ipl->upload(we.data(), we.size(), &ipl->d_we);
        if (rc == SNNHIP_OK) rc = ipl->upload(wp.data(), wp.size(), &ipl->d_wp);
        if (rc == SNNHIP_OK) rc = ipl->upload(e3p.data(), e3p.size(), &ipl->d_e3);
        if (rc == SNNHIP_OK) rc = ipl->upload(reinterpret_cast<const float*>(imgTabs.data()), imgTabs.size(), &ipl->d_tabs);
        if (rc != SNNHIP_OK) {
            delete ipl;
            return rc;
        }
        memcpy(ipl->inDims, expandPlan->inDims, sizeof(ipl->inDims));
        memcpy(ipl->outDims, projectPlan->outDims, sizeof(ipl->outDims));
        ipl->flops = ce->flops + cd->flops + cp->flops;
        ipl->bytes = ce->bytes + cd->bytes + cp->bytes + (addPlan ? addPlan->bytes : 0.0); // unfused accounting of the layers it replaces
        ipl->kernelBytes = fusedBytes;
        char ib[400];
        snprintf(ib, sizeof(ib), "irb_fused_mfma_%s [conv1x1 %d->%d + depthwise3x3 s%d + conv1x1 %d->%d%s] image per block (%dx%d, %d px tiles%s), hidden quarter per wave, "
                 "slices=%d lds=%zuB hbm_bytes=%.6g kernel=irb_image_kernel<%d,%d,%d,true,%s> relu6-epilogues",
                 imgS16 ? "f16x3split_16x16x32" : "f32_16x16x4", C, Ch, s, Ch, Co, addPlan ? " + add" : "", ip.H, ip.W, ip.MT, imgOS > 1 ? ", two blocks of half the output channels" : "",
                 Ch / 16, imgLds, fusedBytes, Co / 16 / imgOS, C / 16, static_cast<int>(imgTabs.size() / 16) - ip.MT, imgS16 ? "true" : "false");
        ipl->desc = ib;
        *out = ipl;
        return SNNHIP_OK;
    }
    auto* plan = new IrbPlan();
    plan->ctx = ctx;
    plan->p = p;
    plan->kernel = fn;
    plan->threads = 64 * NWv;
    plan->ldsBytes = lds;
    plan->grid = dim3(static_cast<unsigned>(up_div(p.tilesX * p.tilesY * p.N, NWv)));
    plan->dtype = SNNHIP_F32;
    int rc = plan->upload(we.data(), we.size(), &plan->d_we);
    if (rc == SNNHIP_OK) rc = plan->upload(wp.data(), wp.size(), &plan->d_wp);
    if (rc == SNNHIP_OK) rc = plan->upload(e3p.data(), e3p.size(), &plan->d_e3);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    memcpy(plan->inDims, cs ? stemPlan->inDims : noExpand ? dwPlan->inDims : expandPlan->inDims, sizeof(plan->inDims));
    memcpy(plan->outDims, projectPlan->outDims, sizeof(plan->outDims));
    plan->flops = (ce ? ce->flops : 0.0) + (cs ? cs->flops : 0.0) + cd->flops + cp->flops;
    plan->bytes = (ce ? ce->bytes : 0.0) + (cs ? cs->bytes : 0.0) + cd->bytes + cp->bytes + (addPlan ? addPlan->bytes : 0.0); // unfused accounting of the layers it replaces
    plan->kernelBytes = fusedBytes;
    if (snnhip::option("SNNHIP_IRB_OCC")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(fn), 64 * NWv, lds);
        fprintf(stderr, "[irb occ] wave G=%d NW=%d lds=%zu grid=%u: %d blocks per CU\n", G, NWv, lds, plan->grid.x, nb);
    }
    char buf[320];
    char head[64];
    if (cs) snprintf(head, sizeof(head), "stem conv3x3 s%d 3->%d + depthwise3x3 s%d", p.stemS, Ch, s);
    else if (noExpand) snprintf(head, sizeof(head), "depthwise3x3 %d s%d", Ch, s);
    else snprintf(head, sizeof(head), "conv1x1 %d->%d + depthwise3x3 s%d", C, Ch, s);
    // the instantiation pick_irb_wave chose, for the profile look-up (bench.py matches the PMC record of exactly this kernel)
    int vNcb = 0, vCj = 0;
    {
        static const int kVariants[][2] = {{2, 1}, {2, 2}, {4, 2}, {4, 4}, {6, 4}, {6, 6}, {10, 6}, {10, 10}, {20, 10}};
        for (const auto& v : kVariants)
            if (p.NCB <= v[0] && p.Cj <= v[1]) {
                vNcb = v[0];
                vCj = v[1];
                break;
            }
    }
    snprintf(buf, sizeof(buf), "irb_fused_mfma_%s [%s + conv1x1 %d->%d%s] tile=%dx8px per wave, halo=%dx%d slices=%d threads=%d lds=%zuB hbm_bytes=%.6g kernel=irb_wave_kernel<%d,%d,%d,%s,%s,%s>",
             waveS16 ? "f16x3split_16x16x32" : "f32_16x16x4", head, Ch, Co, addPlan ? " + add" : "", 2 * G, p.HH, p.HWd, p.nChunks, plan->threads, lds, fusedBytes, G, cs ? 2 : vNcb,
             cs ? 2 : vCj, r6 ? "true" : "false", cs ? "true" : "false", waveS16 ? "true" : "false");
    plan->desc = buf;
    if (r6) plan->desc += " relu6-epilogues";
    if (tail8) plan->desc += " tail8";
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
