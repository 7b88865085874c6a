// conv2d_mfma_kernel.h -- the implicit-GEMM convolution kernel template of conv2d_mfma.hip and its variant table.  It lives in a header so that
// the ~170 instantiations can be compiled as three translation units (conv2d_mfma_bn{128,64,32}_{f32,f16}.hip, one per block width and precision) side by side: as one
// unit they took 3.5 minutes of a 4-minute build.  See conv2d_mfma.hip for the design notes.
#pragma once
#include "epilogue.h"
#include "snnhip_internal.h"

#include <cmath>
#include <cstdlib>
#include <map>
#include <type_traits>

#ifndef SNNHIP_ABL
#define SNNHIP_ABL 0 // ablation builds only, see tools/ablate_conv.sh
#endif

namespace snnhip {

namespace mfma_detail {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MfmaParams { // (declared after ActCfg: epilogue.h)
    int N, H, W, IC, OC, kh, kw, sh, sw, padx, pady, padMode, useBN, OH, OW;
    int TBs, THs, TWs;   // log2 of the pixel-tile dims
    int tileH, tileW;    // staged input tile (per image of the tile)
    int rowPitch;        // LDS pixels per staged row  (sh*rowPitch == TW mod 16 -> conflict-free ds_read_b128, see lds_off)
    int imgPitch;        // LDS pixels per staged image (multiple of 16)
    int evenCols;        // sw == 2: columns are stored de-interleaved, [even columns | odd columns]; else 0
    int tilesX, tilesY;  // pixel tiles along x / y (tiles along batch = gridDim.x / (tilesX*tilesY))
    int nChunks;         // ceil(IC / ICc)
    int OCp;             // OC padded to a multiple of the block's BN
    int total;           // float4 elements staged per chunk
    int bufFloats;       // floats per LDS buffer
    int splitK;          // > 1: blockIdx.z owns chunks [z*chunksPerSplit, ...) and stores raw partial sums to the workspace
    int chunksPerSplit;
    int ldsEpi;          // fp16, OC % 8 == 0, no split-K: the output tile leaves through LDS as 16-byte channel-contiguous stores
    int preMode, preX, preY, srcH, srcW; // fused Pad layer (ConvGeom): H, W are the padded dims, the tensor is srcH x srcW (== H, W when preMode == 0)
    int preShift;                        // fused nearest x2 upsampling in front of the pad: resolve against (srcH, srcW) << 1, then >> 1
    unsigned magicW, magicH; // ceil(2^32 / tileW), ceil(2^32 / tileH): the prologue's divisions by run-time values become one v_mul_hi each
    // fused residual Add (chain rule E): y = act2(conv_result + res), res = a tensor of the output's shape and type, set per launch
    const void* res;
    ActCfg ac2;
    // per-tile output statistics for a following InstanceNorm (chain rule F; fp16 LDS epilogue, one image per pixel tile, no split-K):
    // statPart[((n*tilesY + ty)*tilesX + tx)][2][OC] = {mean, sum of squared deviations} of the tile's valid pixels, per channel
    float* statPart;
    // InstanceNorm in front of the convolution (graph rule I; fp16 kernels, one image per pixel tile, IC % 8 == 0): applied to the staged values
    const float* normShift; // y = normAc(x * mul[n][c] + shift[n][c]): the norm's own fma (eltwise_pool.hip)
    const float* normMul;
    ActCfg normAc;
    int normTabOfs; // float offset in LDS of [shift | mul] x IC of the tile's image
    int coordTabOfs; // float offset in LDS of the staging's row / column tables (tileH + tileW ints)
};

// LDS layout of the staged activations.  A pixel owns ICc floats = ICc/4 16-byte slots; the slot is XOR-swizzled with
// bits of the linear pixel index so that the 16-byte bank slot (address/16 mod 16) is a bijection of (pixel mod 16):
// a ds_read_b128 lane group (16 lanes: {0-3,12-15,20-27} / {4-11,16-19,28-31} of each half-wave, MI355X_MICROARCH.md
// section LDS) is conflict-free iff its 16 pixels are distinct mod 16, which the row/image pitches guarantee.
template <int C8>
__device__ __forceinline__ int lds_off(int pl, int slot) {
    // Q = 2*C8 slots per pixel (2, 4, 8 or 16): slot ^ ((pl >> (4 - log2 Q)) & (Q-1)) makes (address / 16) mod 16 a bijection of pl mod 16
    if (C8 == 0) return pl * 4; // tap-pair mode: one 16-byte slot per pixel, consecutive pixels are consecutive bank slots
    constexpr int Q = 2 * C8;
    constexpr int LQ = Q == 2 ? 1 : Q == 4 ? 2 : Q == 8 ? 3 : 4;
    return pl * (4 * Q) + ((slot ^ ((pl >> (4 - LQ)) & (Q - 1))) << 2);
}

// F16: tensors and packed weights hold halfs; a 16-byte slot is 8 channels instead of 4 and ONE v_mfma_f32_32x32x16_f16 consumes the slot pair
// (h = 0, 1) that four v_mfma_f32_32x32x2_f32 consume in fp32, so a K-step is 16 channels; byte geometry (LDS tile, swizzle, weight stream,
// 128-bit operand loads) is identical.  Accumulation and epilogue stay fp32; stores round to nearest even.
//
// TAPS > 0 (fp16 only): the kernel-tap count is a compile-time constant and the K loop of a chunk (S = TAPS * C8 steps) is fully unrolled, so
// the weight ring is addressed with static indices (no register shuffling) and can be D = 6..9 steps deep.  An fp16 K step is 4 MFMAs of
// 32 cycles for a 2x2 register block -- the 2-step ring of the rolled loop (right for fp32, whose step is 8x longer) left the wave waiting
// for L2 on every tap (s_waitcnt vmcnt(0) at the loop head, 32 % MFMA utilisation on the U-Net / ResNet 3x3 layers).
//
// C8 == 0, "tap-pair" mode for channel-thin inputs (IC <= 4 fp32 / <= 8 fp16: the RGB stems of ResNet / MobileNetV2 / YOLO / Candy): a pixel
// is ONE 16-byte slot and the K axis runs over the taps instead of the channels -- the two lane halves of a K step read two DIFFERENT taps
// (h = 0: tap 2j, h = 1: tap 2j+1) of the same pixel slot layout, so a 7x7x3 stem takes 25 K steps instead of 49 and 3 of every 4 (fp32)
// operand lanes carry data instead of 3 of 8.  Weights are packed [step][h][oc] to match; an odd tap count pads the last h = 1 half with zeros.
template <int WM, int WN, int MT, int NT, int C8, int R, bool SIMPLE, bool F16, int TAPS = 0>
__global__ __launch_bounds__(256, 2) void conv2d_mfma_kernel(MfmaParams p, ActCfg ac, const void* __restrict__ xv, const void* __restrict__ wpv,
                                                          const float4* __restrict__ epi, void* __restrict__ yv, float* __restrict__ ws) {
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(WM * MT == 4, "128 pixels per block");
    typedef typename std::conditional<F16, _Float16, float>::type T;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    constexpr int CH = F16 ? 8 : 4; // channels per 16-byte slot
    const T* __restrict__ x = static_cast<const T*>(xv);
    const float4* __restrict__ wp = static_cast<const float4*>(wpv);
    T* __restrict__ y = static_cast<T*>(yv);
    constexpr int Q = C8 ? 2 * C8 : 1; // 16-byte slots per staged pixel
    constexpr bool PAIR = C8 == 0;
    constexpr int BN = 32 * NT * WN;
    constexpr int S = TAPS * C8; // K steps per chunk when the tap count is static
    constexpr int DS = S % 6 == 0 ? 6 : (S == 9 ? 9 : (S == 8 ? 8 : (S % 4 == 0 ? 4 : (S % 3 == 0 ? 3 : (S % 2 == 0 ? 2 : 1))))); // divides S
    constexpr int D = TAPS ? DS : ((MT * NT >= 4) ? 2 : (MT * NT == 2 ? 3 : 4)); // weight prefetch distance in K steps
    static_assert(!TAPS || F16, "the unrolled K loop is instantiated for fp16 only");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int l32 = lane & 31, h = lane >> 5;

    const int mt = blockIdx.x;
    const int tx = mt % p.tilesX, ty = (mt / p.tilesX) % p.tilesY, tb = mt / (p.tilesX * p.tilesY);
    const int TWm = (1 << p.TWs) - 1, THm = (1 << p.THs) - 1;
    const int ox0 = tx << p.TWs, oy0 = ty << p.THs, b0 = tb << p.TBs;
    const int ix0 = ox0 * p.sw - p.padx, iy0 = oy0 * p.sh - p.pady;
    const int taps = p.kh * p.kw;
    const bool vec4 = (p.IC % CH) == 0;

    // ---- the source pixel of a staged pixel is separable (its row depends on the tile row only, its column on the tile column only): padding /
    // fused Pad / fused UpSampling are resolved once per tile row and column into two LDS tables, an element below is two look-ups (resolved per
    // element it was ~60 of the ~100 instructions an element costs, x R elements, in blocks whose waves run 36-72 MFMAs)
    int* const syTab = reinterpret_cast<int*>(smem + p.coordTabOfs); // [tileH] first pixel of the source row (row * srcW), -1 = outside (zeros)
    int* const sxTab = syTab + p.tileH;                               // [tileW] source column, -1 = outside
    for (int i = tid; i < p.tileH + p.tileW; i += 256) {
        const bool isRow = i < p.tileH;
        int s1 = resolve_nobranch(isRow ? iy0 + i : ix0 + (i - p.tileH), isRow ? p.H : p.W, p.padMode);
        if (p.preMode) { // (uniform) a pixel of the (virtual) padded image -> the source pixel the Pad layer would have copied; -1 stays -1
            const int pp = resolve_nobranch(s1 - (isRow ? p.preY : p.preX), (isRow ? p.srcH : p.srcW) << p.preShift, p.preMode);
            s1 = s1 < 0 ? -1 : (pp < 0 ? -1 : pp >> p.preShift); // nearest x2: upsampled pixel (y, x) is source pixel (y / 2, x / 2) (vk_upsampling2d_nearest.comp:50-65)
        }
        syTab[i] = (isRow && s1 >= 0) ? s1 * p.srcW : s1;
    }
    __syncthreads();

    // ---- staging descriptors: element e = tid + 256 r -> (pixel of the halo tile, channel quad q); q is the same for all r
    const int q = tid & (Q - 1);
    int gofs[R], lofs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = tid + 256 * r;
        gofs[r] = -1;
        lofs[r] = -1;
        if (e < p.total) {
            const int pix = e / Q;
            // exact for pix < 2^16 and divisors < 2^16 (pix <= 9 * 256); a divisor of 1 has no 32-bit magic number
            const int t2 = p.tileW == 1 ? pix : static_cast<int>(__umulhi(static_cast<unsigned>(pix), p.magicW));
            const int c = pix - t2 * p.tileW;
            const int b = p.tileH == 1 ? t2 : static_cast<int>(__umulhi(static_cast<unsigned>(t2), p.magicH));
            const int rr = t2 - b * p.tileH;
            const int rowPix = syTab[rr], sx = sxTab[c];
            const int n = b0 + b;
            const int cm = p.evenCols ? (c & 1) * p.evenCols + (c >> 1) : c;
            lofs[r] = lds_off<C8>(b * p.imgPitch + rr * p.rowPitch + cm, q);
            if (rowPix >= 0 && sx >= 0 && n < p.N) gofs[r] = (n * p.srcH * p.srcW + rowPix + sx) * p.IC + q * CH;
        }
    }
    float4 stage[R];
    auto stage_load = [&](int ic0) {
        const int icq = ic0 + q * CH;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gofs[r] >= 0 && icq < p.IC && !(SNNHIP_ABL & 4)) { // ablation bit 4: no activation loads
                const T* src = x + gofs[r] + ic0;
                if (vec4) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    T tmp[CH];
#pragma unroll
                    for (int j = 0; j < CH; ++j) tmp[j] = icq + j < p.IC ? src[j] : static_cast<T>(0.0f);
                    v = *reinterpret_cast<const float4*>(tmp);
                }
            }
            stage[r] = v;
        }
    };
    auto stage_store = [&](float* buf, int ic0) {
        if constexpr (F16) {
            if (p.normShift && ic0 + q * CH < p.IC) { // block-uniform switch; a thread's elements share their 8 channels
                const float* tab = smem + p.normTabOfs + ic0 + q * CH;
                float sh[8], mul[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    sh[k] = tab[k];
                    mul[k] = tab[p.IC + k];
                }
                const bool relu = p.normAc.act == SNNHIP_ACT_RELU; // (uniform) one instruction instead of the general mul / max / med3
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (gofs[r] >= 0) { // zero padding stays zero: the border applies to the NORMALISED tensor
                        h8 hv = *reinterpret_cast<const h8*>(&stage[r]);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float v = fmaf(static_cast<float>(hv[k]), mul[k], sh[k]);
                            hv[k] = static_cast<_Float16>(relu ? fmaxf(v, 0.0f) : __builtin_amdgcn_fmed3f(fmaxf(v, v * p.normAc.alpha), p.normAc.lo, p.normAc.hi));
                        }
                        stage[r] = *reinterpret_cast<const float4*>(&hv);
                    }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (lofs[r] >= 0) *reinterpret_cast<float4*>(buf + lofs[r]) = stage[r];
    };

    // ---- MFMA operand addressing: lane (l32, h) reads pixel i = subtile*32 + l32, channels 4h..4h+3 of each 8-channel step
    int apix[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int i = (wm * MT + t) * 32 + l32;
        const int b = i >> (p.THs + p.TWs), py = (i >> p.TWs) & THm, px = i & TWm;
        apix[t] = b * p.imgPitch + py * p.sh * p.rowPitch + (p.evenCols ? px : px * p.sw); // sw==2: column 2px+fx -> plane (fx&1), index px+(fx>>1)
    }
    const int n0 = blockIdx.y * BN + wn * (NT * 32);
    const size_t bstep = static_cast<size_t>(2) * p.OCp; // float4 (16-byte) units per K step
    const int chunk0 = blockIdx.z * p.chunksPerSplit, chunk1 = min(p.nChunks, chunk0 + p.chunksPerSplit);
    const float4* bptr = wp + (static_cast<size_t>(h) * p.OCp + n0 + l32) + static_cast<size_t>(chunk0) * taps * C8 * bstep;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    // one K step: fp16 = one v_mfma_f32_32x32x16_f16 per (t, u) register tile (16 channels: both lane halves' 8 halfs); fp32 = four
    // v_mfma_f32_32x32x2_f32, one per component of the 16-byte operands (8 channels)
    auto k_step = [&](const float4 (&a)[MT], const float4 (&b)[NT]) {
        if (F16) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&a[t]), *reinterpret_cast<const h8*>(&b[u]), acc[t][u], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[u].x, acc[t][u], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[u].y, acc[t][u], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b[u].z, acc[t][u], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[u].w, acc[t][u], 0, 0, 0);
        }
    };

    int tapDelta[TAPS > 0 ? TAPS : 1]; // LDS pixel delta of each tap (static tap count only)
    if constexpr (TAPS > 0) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int fy = t / p.kw, fxx = t - fy * p.kw;
            tapDelta[t] = fy * p.rowPitch + (p.evenCols ? (fxx & 1) * p.evenCols + (fxx >> 1) : fxx);
        }
    }

    // weight ring: bq[d] = step s+d (the packed array carries D extra zero steps at the end)
    float4 bq[D][NT];
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int u = 0; u < NT; ++u) bq[d][u] = bptr[u * 32];
        bptr += bstep;
    }

    stage_load(chunk0 * 2 * CH * C8);
    if constexpr (F16) {
        if (p.normShift) { // the norm's (shift, multiplier) of this tile's image (TB == 1) -> LDS
            float* tab = smem + p.normTabOfs;
            const int n = min(b0, p.N - 1);
            for (int i = tid; i < p.IC; i += 256) {
                tab[i] = p.normShift[static_cast<size_t>(n) * p.IC + i];
                tab[p.IC + i] = p.normMul[static_cast<size_t>(n) * p.IC + i];
            }
            __syncthreads();
        }
    }
    stage_store(smem, chunk0 * 2 * CH * C8); // buffer parity is relative to the split's first chunk: a single-chunk split needs one buffer only
    __syncthreads();

    for (int chunk = chunk0; chunk < chunk1; ++chunk) {
        const float* cur = smem + ((chunk - chunk0) & 1) * p.bufFloats;
        const bool more = chunk + 1 < chunk1;
        if (more) stage_load((chunk + 1) * 2 * CH * C8);

        float4 an[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) an[t] = *reinterpret_cast<const float4*>(cur + lds_off<C8>(apix[t], h));
        if constexpr (TAPS > 0) {
            // S = TAPS * C8 steps, fully unrolled; D divides S, so step s of every chunk lives in ring slot s % D
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float4 a[MT], b[NT];
#pragma unroll
                for (int t = 0; t < MT; ++t) a[t] = an[t];
#pragma unroll
                for (int u = 0; u < NT; ++u) b[u] = bq[s % D][u];
#if !(SNNHIP_ABL & 1) // ablation builds (tools/ablate_conv.sh): 1 = no weight refills, 2 = no LDS operand reads
#pragma unroll
                for (int u = 0; u < NT; ++u) bq[s % D][u] = bptr[u * 32];
                bptr += bstep;
#endif
                __builtin_amdgcn_sched_barrier(0); // keep the refill D steps ahead of its use: the scheduler otherwise sinks it next to the consumer
                if (s + 1 < S && !(SNNHIP_ABL & 2)) {
                    const int dl = tapDelta[(s + 1) / C8];
                    const int slot = ((s + 1) % C8) * 2 + h;
#pragma unroll
                    for (int t = 0; t < MT; ++t) an[t] = *reinterpret_cast<const float4*>(cur + lds_off<C8>(apix[t] + dl, slot));
                }
                k_step(a, b);
            }
            if (more) stage_store(smem + ((chunk + 1 - chunk0) & 1) * p.bufFloats, (chunk + 1) * 2 * CH * C8);
            __syncthreads();
            continue;
        }
        if constexpr (PAIR) {
            // this lane walks taps h, h+2, h+4, ...: (pfx, prow) = column and LDS row offset of its current tap
            const int steps = (taps + 1) >> 1;
            auto tap_delta = [&](int fxx, int row) { return row + (p.evenCols ? (fxx & 1) * p.evenCols + (fxx >> 1) : fxx); };
            int ptap = h, pfx = h % p.kw, prow = (h / p.kw) * p.rowPitch;
            if (ptap >= taps) pfx = prow = 0; // 1x1 never comes here (taps >= 2), an odd tap count ends on a zero-weight half: read tap 0
#pragma unroll
            for (int t = 0; t < MT; ++t) an[t] = *reinterpret_cast<const float4*>(cur + lds_off<0>(apix[t] + tap_delta(pfx, prow), 0));
#pragma unroll 1
            for (int j = 0; j < steps; ++j) {
                float4 a[MT], b[NT];
#pragma unroll
                for (int t = 0; t < MT; ++t) a[t] = an[t];
#pragma unroll
                for (int u = 0; u < NT; ++u) b[u] = bq[0][u];
#pragma unroll
                for (int d = 0; d + 1 < D; ++d)
#pragma unroll
                    for (int u = 0; u < NT; ++u) bq[d][u] = bq[d + 1][u];
#pragma unroll
                for (int u = 0; u < NT; ++u) bq[D - 1][u] = bptr[u * 32];
                bptr += bstep;
                ptap += 2;
                pfx += 2;
                while (pfx >= p.kw) {
                    pfx -= p.kw;
                    prow += p.rowPitch;
                }
                if (ptap >= taps) pfx = prow = 0;
#pragma unroll
                for (int t = 0; t < MT; ++t) an[t] = *reinterpret_cast<const float4*>(cur + lds_off<0>(apix[t] + tap_delta(pfx, prow), 0));
                k_step(a, b);
            }
            if (more) stage_store(smem + ((chunk + 1 - chunk0) & 1) * p.bufFloats, (chunk + 1) * 2 * CH * C8);
            __syncthreads();
            continue;
        }
        int fx = 0, rowoff = 0;
#pragma unroll 1
        for (int tap = 0; tap < taps; ++tap) {
            // pixel delta of the NEXT tap (clamped to tap 0 after the last one: that prefetch is never consumed)
            int fxn = fx + 1, rown = rowoff;
            if (fxn == p.kw) {
                fxn = 0;
                rown += p.rowPitch;
            }
            if (tap + 1 == taps) {
                fxn = 0;
                rown = 0;
            }
            const int dcur = rowoff + (p.evenCols ? (fx & 1) * p.evenCols + (fx >> 1) : fx);
            const int dnext = rown + (p.evenCols ? (fxn & 1) * p.evenCols + (fxn >> 1) : fxn);
#pragma unroll
            for (int c8 = 0; c8 < C8; ++c8) {
                float4 a[MT], b[NT];
#pragma unroll
                for (int t = 0; t < MT; ++t) a[t] = an[t];
#pragma unroll
                for (int u = 0; u < NT; ++u) b[u] = bq[0][u];
#pragma unroll
                for (int d = 0; d + 1 < D; ++d)
#pragma unroll
                    for (int u = 0; u < NT; ++u) bq[d][u] = bq[d + 1][u];
#pragma unroll
                for (int u = 0; u < NT; ++u) bq[D - 1][u] = bptr[u * 32];
                bptr += bstep;
                {
                    const int dl = (c8 + 1 < C8) ? dcur : dnext;
                    const int slot = (c8 + 1 < C8) ? (c8 + 1) * 2 + h : h;
#pragma unroll
                    for (int t = 0; t < MT; ++t) an[t] = *reinterpret_cast<const float4*>(cur + lds_off<C8>(apix[t] + dl, slot));
                }
                k_step(a, b);
            }
            fx = fxn;
            rowoff = rown;
        }
        if (more) stage_store(smem + ((chunk + 1 - chunk0) & 1) * p.bufFloats, (chunk + 1) * 2 * CH * C8);
        __syncthreads();
    }

    // ---- epilogue: bias -> BN -> activation, 128-byte channel-contiguous stores.  Rows r&3 of a lane are 4 adjacent
    // x pixels of one image row (TW >= 4, tile origins multiples of 4) -> one 32-bit offset per group of 4 rows.
    // fp16 (p.ldsEpi): a lane's accumulators are single halfs of 16 different pixels -- stored directly they leave as 2-byte scatters in
    // 64-byte runs (measured: 19 of 69 us of a 3x3 256->128 layer).  The tile is transposed through LDS instead ([pixel][BN halfs], row
    // pitch BN*2+16 bytes so that the two half-waves hit disjoint banks) and written as 16-byte vectors, a pixel's BN channels contiguous.
    constexpr int EPITCH = BN + 8; // halfs per LDS row of the output tile
    const bool addSimple = act_is_simple_dev(p.ac2.act);
    _Float16* const otile = reinterpret_cast<_Float16*>(smem);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int ibase = (wm * MT + t) * 32 + 4 * h;
        // fused Add: this M-tile's 16*NT residual values are requested in one batch before the first store of the tile.  Interleaved with the
        // stores (y may alias res as far as the compiler knows) every load waits out its full latency: the fused layer was SLOWER than
        // convolution + separate Add launch (56x56 64->64 b32: 111 + 12 us apart, 131 us fused).
        float rv[4][NT][4];
        const bool resDirect = p.res && p.splitK == 1 && !(F16 && p.ldsEpi);
        if (resDirect) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = ibase + 8 * g;
                const int b = i >> (p.THs + p.TWs), py = (i >> p.TWs) & THm, px = i & TWm;
                const int n = b0 + b, oy = oy0 + py, ox = ox0 + px;
                const bool rowOk = n < p.N && oy < p.OH;
                const int pofs = ((n * p.OH + oy) * p.OW + ox) * p.OC;
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const int oc = n0 + u * 32 + l32;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        rv[g][u][k] = (rowOk && oc < p.OC && ox + k < p.OW) ? static_cast<float>(static_cast<const T*>(p.res)[pofs + k * p.OC + oc]) : 0.0f;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int i = ibase + 8 * g;
            const int b = i >> (p.THs + p.TWs), py = (i >> p.TWs) & THm, px = i & TWm;
            const int n = b0 + b, oy = oy0 + py, ox = ox0 + px;
            const bool rowOk = n < p.N && oy < p.OH;
            const int pofs = ((n * p.OH + oy) * p.OW + ox) * p.OC;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int oc = n0 + u * 32 + l32;
                const float4 e = epi[oc]; // table padded to OCp
                const bool ok = rowOk && oc < p.OC;
                if (p.splitK > 1) { // ws = fp32 workspace [splitK][N*OH*OW][OC]: raw partial sums, epilogue in splitk_reduce_kernel
                    float* wz = ws + static_cast<size_t>(blockIdx.z) * p.N * p.OH * p.OW * p.OC;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (ok && ox + k < p.OW) wz[pofs + k * p.OC + oc] = acc[t][u][4 * g + k];
                    continue;
                }
                float first = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = epi_affine(acc[t][u][4 * g + k], e, p.useBN);
                    if (SIMPLE) {
                        v = apply_act<true>(ac, v, 0.0f);
                    } else {
                        const int act = (ac.act == SNNHIP_ACT_SILU_QUIRK && k == 0) ? SNNHIP_ACT_SILU : ac.act;
                        v = epi_act(act, ac.leaky, v, first);
                        if (k == 0) first = v;
                    }
                    if (F16 && p.ldsEpi) {
                        otile[(i + k) * EPITCH + wn * (NT * 32) + u * 32 + l32] = static_cast<_Float16>(v);
                    } else if (ok && ox + k < p.OW && (!(SNNHIP_ABL & 8) || v == 12345.678f)) { // ablation bit 8: no output stores
                        if (resDirect) { // the Add layer behind this convolution: same rounding points as the two separate launches
                            const float cv = static_cast<float>(static_cast<T>(v));
                            v = add_act(p.ac2, addSimple, cv + rv[g][u][k]);
                        }
                        y[pofs + k * p.OC + oc] = static_cast<T>(v);
                    }
                }
            }
        }
    }
    if (F16 && p.ldsEpi) {
        __syncthreads();
        constexpr int VPR = BN / 8; // 16-byte vectors per pixel row of the tile
        // chain rule F (p.statPart, block-uniform): a thread's vectors are 8 pixels of ONE 8-channel column (256 % VPR == 0), so the sums and
        // sums of squares of the stored (rounded) values accumulate in 16 registers while the vectors pass through
        float sa[8], sb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sa[e] = sb[e] = 0.0f;
#pragma unroll
        for (int j = 0; j < 128 * VPR / 256; ++j) {
            const int v = tid + 256 * j;
            const int i = v / VPR, c8 = v - i * VPR;
            const int b = i >> (p.THs + p.TWs), py = (i >> p.TWs) & THm, px = i & TWm;
            const int n = b0 + b, oy = oy0 + py, ox = ox0 + px;
            const int oc = blockIdx.y * BN + c8 * 8;
            if (n < p.N && oy < p.OH && ox < p.OW && oc < p.OC && !(SNNHIP_ABL & 8)) {
                const size_t o = static_cast<size_t>((n * p.OH + oy) * p.OW + ox) * p.OC + oc;
                float4 pack = *reinterpret_cast<const float4*>(otile + i * EPITCH + c8 * 8);
                if (p.statPart) {
                    const _Float16* ch = reinterpret_cast<const _Float16*>(&pack);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = static_cast<float>(ch[e]);
                        sa[e] += f;
                        sb[e] = fmaf(f, f, sb[e]);
                    }
                }
                if (p.res) {
                    const float4 rpack = *reinterpret_cast<const float4*>(static_cast<const T*>(p.res) + o);
                    const _Float16* ch = reinterpret_cast<const _Float16*>(&pack);
                    const _Float16* rh = reinterpret_cast<const _Float16*>(&rpack);
                    _Float16 oh[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) oh[e] = static_cast<_Float16>(add_act(p.ac2, addSimple, static_cast<float>(ch[e]) + static_cast<float>(rh[e])));
                    pack = *reinterpret_cast<const float4*>(oh);
                }
                *reinterpret_cast<float4*>(y + o) = pack;
            }
        }
        if (p.statPart) { // fold the 256 / VPR pixel groups per channel in a fixed order (the tile in LDS is dead by now: its space is the scratch)
            constexpr int PG = 256 / VPR;
            float* const sred = reinterpret_cast<float*>(smem);
            __syncthreads();
            {
                const int c8 = tid % VPR, pg = tid / VPR;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sred[pg * BN + c8 * 8 + e] = sa[e];
                    sred[PG * BN + pg * BN + c8 * 8 + e] = sb[e];
                }
            }
            __syncthreads();
            const int ocb = blockIdx.y * BN;
            if (tid < BN && ocb + tid < p.OC) {
                float a1 = 0.0f, a2 = 0.0f;
                for (int j = 0; j < PG; ++j) {
                    a1 += sred[j * BN + tid];
                    a2 += sred[PG * BN + j * BN + tid];
                }
                const float an = static_cast<float>(min(1 << p.THs, p.OH - oy0) * min(1 << p.TWs, p.OW - ox0)); // valid pixels of this tile (TB == 1)
                float* po = p.statPart + (static_cast<size_t>(b0 * p.tilesY + (oy0 >> p.THs)) * p.tilesX + (ox0 >> p.TWs)) * 2 * p.OC;
                po[ocb + tid] = a1 / an;
                po[p.OC + ocb + tid] = fmaxf(a2 - a1 * a1 / an, 0.0f);
            }
        }
    }
}

// split-K second pass: y[m][oc] = act(BN(bias + sum_z ws[z][m][oc])), summed in a fixed order (deterministic)
typedef void (*KernelFn)(MfmaParams, ActCfg, const void*, const void*, const float4*, void*, float*);

// the variant table, one precision per instantiation (F16 is a template parameter so that a translation unit compiles one precision only)
template <int WM, int WN, int MT, int NT, bool F16>
KernelFn pick_kernel(int c8, int r, bool simple, int taps) {
    if constexpr (F16) {
        // static tap count: 3x3 (and the 2x2 of U-Net's up-convolutions) with 16/32-channel chunks
#define SNNHIP_PICK_T(C8_, R_, T_)                                                                                                          \
    if (taps == T_ && c8 == C8_ && r == R_)                                                                                                   \
        return simple ? conv2d_mfma_kernel<WM, WN, MT, NT, C8_, R_, true, true, T_> : conv2d_mfma_kernel<WM, WN, MT, NT, C8_, R_, false, true, T_>;
        if (!snnhip::option("SNNHIP_CONV_ROLLED")) {
            SNNHIP_PICK_T(1, 3, 9)
            SNNHIP_PICK_T(2, 3, 9)
            SNNHIP_PICK_T(2, 5, 9)
            SNNHIP_PICK_T(2, 3, 4)
        }
#undef SNNHIP_PICK_T
    }
#define SNNHIP_PICK(C8_, R_) \
    if (c8 == C8_ && r == R_) return simple ? conv2d_mfma_kernel<WM, WN, MT, NT, C8_, R_, true, F16> : conv2d_mfma_kernel<WM, WN, MT, NT, C8_, R_, false, F16>;
    SNNHIP_PICK(0, 3)
    SNNHIP_PICK(0, 5)
    SNNHIP_PICK(1, 3)
    SNNHIP_PICK(1, 5)
    SNNHIP_PICK(1, 9)
    SNNHIP_PICK(2, 3)
    SNNHIP_PICK(2, 5)
    SNNHIP_PICK(2, 9)
    SNNHIP_PICK(4, 5)
    SNNHIP_PICK(4, 9)
    SNNHIP_PICK(8, 9)
#undef SNNHIP_PICK
    return nullptr;
}

} // namespace mfma_detail
} // namespace snnhip
