// conv2d_mfma.hip -- fp32 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32) for the
// GEMM-shaped layers of the reference's Conv2D operator (3x3 / 1x1 / 7x7 ... with IC >= 8): ResNet-18 / MobileNetV2
// pointwise / Candy style convolutions (BASELINE configs[2..4]).
//
// Replaces shadertemplate_vk_conv2d.comp:148-347 and shadertemplate_vk_conv2d_1x1.comp:68-210 of the reference for those
// shapes; arithmetic, padding modes, epilogue (bias -> BN -> activation) and the SiLU 4-pixel quirk are the same as in
// conv2d_generic.hip, only the reduction order differs (fp32 accumulate in the MFMA).
//
// GEMM view:  M = output pixels, N = output channels, K = (tap, ic).
//   block  = 256 threads = 4 waves; block tile = 128 pixels x BN channels, BN in {32, 64, 128}
//   pixels of a block tile = TB images x TH rows x TW columns (TB*TH*TW = 128, picked per layer to fit OH x OW)
//   wave   = MT x NT MFMA tiles of 32 x 32 (16 fp32 accumulators per lane per tile)
//   A (activations): per 16-channel chunk (8 when IC <= 8) the input halo tile of the block is staged ONCE in LDS
//       lds[(b, row, col)][pitch = chunk + 4]          (pitch 20 / 12 floats: conflict-free ds_read_b128)
//     and re-used by all kh*kw taps: a tap is just a different LDS offset.  Next chunk's tile is fetched into registers
//     while the MFMAs of this chunk run and written to the other LDS buffer afterwards (one barrier per chunk).
//   B (weights): pre-packed on the host so that one lane reads ONE float4 per MFMA tile and K-step, straight from
//     global/L2 (the stream is purely linear over the K loop):
//       Wp[chunk][tap][c8][h][ocPadded][j],   ic = chunk*ICc + c8*8 + h*4 + j
//     The K order inside an 8-channel step is permuted (h = lane/32 owns channels 4h..4h+3, MFMA j consumes component
//     j) so that both operands of four consecutive MFMAs come from one 16-byte load.
//   D: lane l holds column (oc) l%32 and rows 8*(r/4) + 4*(l/32) + r%4 -> a store instruction writes 32 consecutive
//     channels of one pixel (128 B) per half-wave; the epilogue parameters are per-lane constants.
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

namespace {

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
            int sy = resolve_coord(iy0 + rr, p.H, p.padMode);
            int sx = resolve_coord(ix0 + c, p.W, p.padMode);
            if (p.preMode && sy >= 0 && sx >= 0) { // a pixel of the (virtual) padded image -> the source pixel the Pad layer would have copied
                sy = resolve_coord(sy - p.preY, p.srcH << p.preShift, p.preMode);
                sx = resolve_coord(sx - p.preX, p.srcW << p.preShift, p.preMode);
                if (sy >= 0) sy >>= p.preShift; // nearest x2: upsampled pixel (y, x) is source pixel (y / 2, x / 2) (vk_upsampling2d_nearest.comp:50-65)
                if (sx >= 0) sx >>= p.preShift;
            }
            const int n = b0 + b;
            const int cm = p.evenCols ? (c & 1) * p.evenCols + (c >> 1) : c;
            lofs[r] = lds_off<C8>(b * p.imgPitch + rr * p.rowPitch + cm, q);
            if (sy >= 0 && sx >= 0 && n < p.N) gofs[r] = ((n * p.srcH + sy) * p.srcW + sx) * p.IC + q * CH;
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
    auto stage_store = [&](float* buf) {
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
    stage_store(smem); // buffer parity is relative to the split's first chunk: a single-chunk split needs one buffer only
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
            if (more) stage_store(smem + ((chunk + 1 - chunk0) & 1) * p.bufFloats);
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
            if (more) stage_store(smem + ((chunk + 1 - chunk0) & 1) * p.bufFloats);
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
        if (more) stage_store(smem + ((chunk + 1 - chunk0) & 1) * p.bufFloats);
        __syncthreads();
    }

    // ---- epilogue: bias -> BN -> activation, 128-byte channel-contiguous stores.  Rows r&3 of a lane are 4 adjacent
    // x pixels of one image row (TW >= 4, tile origins multiples of 4) -> one 32-bit offset per group of 4 rows.
    // fp16 (p.ldsEpi): a lane's accumulators are single halfs of 16 different pixels -- stored directly they leave as 2-byte scatters in
    // 64-byte runs (measured: 19 of 69 us of a 3x3 256->128 layer).  The tile is transposed through LDS instead ([pixel][BN halfs], row
    // pitch BN*2+16 bytes so that the two half-waves hit disjoint banks) and written as 16-byte vectors, a pixel's BN channels contiguous.
    constexpr int EPITCH = BN + 8; // halfs per LDS row of the output tile
    _Float16* const otile = reinterpret_cast<_Float16*>(smem);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int ibase = (wm * MT + t) * 32 + 4 * h;
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
                        if (p.res) { // the Add layer behind this convolution: same rounding points as the two separate launches
                            const float cv = static_cast<float>(static_cast<T>(v));
                            v = epi_act(p.ac2.act, p.ac2.leaky, cv + static_cast<float>(static_cast<const T*>(p.res)[pofs + k * p.OC + oc]), 0.0f);
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
                if (p.res) {
                    const float4 rpack = *reinterpret_cast<const float4*>(static_cast<const T*>(p.res) + o);
                    const _Float16* ch = reinterpret_cast<const _Float16*>(&pack);
                    const _Float16* rh = reinterpret_cast<const _Float16*>(&rpack);
                    _Float16 oh[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) oh[e] = static_cast<_Float16>(epi_act(p.ac2.act, p.ac2.leaky, static_cast<float>(ch[e]) + static_cast<float>(rh[e]), 0.0f));
                    pack = *reinterpret_cast<const float4*>(oh);
                }
                *reinterpret_cast<float4*>(y + o) = pack;
            }
        }
    }
}

// split-K second pass: y[m][oc] = act(BN(bias + sum_z ws[z][m][oc])), summed in a fixed order (deterministic)
template <bool SIMPLE, typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(size_t MN, int OC, int splitK, int useBN, ActCfg ac, const float* __restrict__ ws,
                                                           const float4* __restrict__ epi, T* __restrict__ y, const T* __restrict__ res, ActCfg ac2) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < MN; i += static_cast<size_t>(gridDim.x) * 256) {
        float v = 0.0f;
        for (int z = 0; z < splitK; ++z) v += ws[static_cast<size_t>(z) * MN + i];
        v = epi_affine(v, epi[i % OC], useBN);
        v = SIMPLE ? apply_act<true>(ac, v, 0.0f) : epi_act(ac.act, ac.leaky, v, 0.0f);
        if (res) v = epi_act(ac2.act, ac2.leaky, static_cast<float>(static_cast<T>(v)) + static_cast<float>(res[i]), 0.0f); // fused residual Add
        y[i] = static_cast<T>(v);
    }
}

struct MfmaConvPlan : ConvPlanBase {
    float* d_ws = nullptr; // split-K workspace
    MfmaParams p;
    ActCfg ac;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    size_t ldsBytes = 0;
    dim3 grid;
    void (*kernel)(MfmaParams, ActCfg, const void*, const void*, const float4*, void*, float*) = nullptr;

    bool fusedAdd = false; // chain rule E: run(x, residual) -> act2(conv(x) + residual)
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == (fusedAdd ? 2 : 1), "conv2d: expects %d input(s), got %d", fusedAdd ? 2 : 1, nIn);
        const snnhip_tensor* x = in[0];
        MfmaParams p = this->p; // per-launch copy: the residual pointer travels in the kernel argument block
        p.res = nullptr;
        if (fusedAdd) {
            const snnhip_tensor* r = in[1];
            SNNHIP_REQUIRE(r->n == p.N && r->h == p.OH && r->w == p.OW && r->c == p.OC && r->dtype == dtype,
                           "conv2d+add: residual %dx%dx%dx%d (dtype %d) does not match the output %dx%dx%dx%d", r->n, r->h, r->w, r->c, r->dtype, p.N, p.OH,
                           p.OW, p.OC);
            p.res = p.splitK > 1 ? nullptr : static_cast<const void*>(r->data); // split-K: the reduce pass adds it
        }
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.srcH && x->w == p.srcW && x->c == p.IC, "conv2d: input dims %dx%dx%dx%d != plan %dx%dx%dx%d",
                       x->n, x->h, x->w, x->c, p.N, p.srcH, p.srcW, p.IC);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.OC, "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d",
                       out->n, out->h, out->w, out->c, p.N, p.OH, p.OW, p.OC);
        hipLaunchKernelGGL(kernel, grid, dim3(256), ldsBytes, ctx->stream, p, ac, static_cast<const void*>(x->data), static_cast<const void*>(d_w),
                           reinterpret_cast<const float4*>(d_epi), static_cast<void*>(out->data), d_ws);
        if (p.splitK > 1) {
            const size_t MN = out->count();
            size_t blocks = (MN + 255) / 256;
            const size_t cap = static_cast<size_t>(ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256) * 8;
            if (blocks > cap) blocks = cap;
            const dim3 gr(static_cast<unsigned>(blocks));
            const float4* e4 = reinterpret_cast<const float4*>(d_epi);
            const bool simple = act_is_simple(ac.act);
            if (dtype == SNNHIP_F16) {
                _Float16* yo = reinterpret_cast<_Float16*>(out->data);
                const _Float16* rr = fusedAdd ? reinterpret_cast<const _Float16*>(in[1]->data) : nullptr;
                if (simple) hipLaunchKernelGGL((splitk_reduce_kernel<true, _Float16>), gr, dim3(256), 0, ctx->stream, MN, p.OC, p.splitK, p.useBN, ac, d_ws, e4, yo, rr, p.ac2);
                else hipLaunchKernelGGL((splitk_reduce_kernel<false, _Float16>), gr, dim3(256), 0, ctx->stream, MN, p.OC, p.splitK, p.useBN, ac, d_ws, e4, yo, rr, p.ac2);
            } else {
                const float* rr = fusedAdd ? in[1]->data : nullptr;
                if (simple) hipLaunchKernelGGL((splitk_reduce_kernel<true, float>), gr, dim3(256), 0, ctx->stream, MN, p.OC, p.splitK, p.useBN, ac, d_ws, e4, out->data, rr, p.ac2);
                else hipLaunchKernelGGL((splitk_reduce_kernel<false, float>), gr, dim3(256), 0, ctx->stream, MN, p.OC, p.splitK, p.useBN, ac, d_ws, e4, out->data, rr, p.ac2);
            }
        }
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

typedef void (*KernelFn)(MfmaParams, ActCfg, const void*, const void*, const float4*, void*, float*);

template <int WM, int WN, int MT, int NT>
KernelFn pick_kernel(int c8, int r, bool simple, bool f16, int taps) {
    // fp16, static tap count: 3x3 (and the 2x2 of U-Net's up-convolutions) with 16/32-channel chunks
#define SNNHIP_PICK_T(C8_, R_, T_)                                                                                                          \
    if (f16 && taps == T_ && c8 == C8_ && r == R_)                                                                                            \
        return simple ? conv2d_mfma_kernel<WM, WN, MT, NT, C8_, R_, true, true, T_> : conv2d_mfma_kernel<WM, WN, MT, NT, C8_, R_, false, true, T_>;
    if (!getenv("SNNHIP_CONV_ROLLED")) {
        SNNHIP_PICK_T(1, 3, 9)
        SNNHIP_PICK_T(2, 3, 9)
        SNNHIP_PICK_T(2, 5, 9)
        SNNHIP_PICK_T(2, 3, 4)
    }
#undef SNNHIP_PICK_T
#define SNNHIP_PICK(C8_, R_)                                                                                                              \
    if (c8 == C8_ && r == R_) {                                                                                                           \
        if (f16) return simple ? conv2d_mfma_kernel<WM, WN, MT, NT, C8_, R_, true, true> : conv2d_mfma_kernel<WM, WN, MT, NT, C8_, R_, false, true>;   \
        return simple ? conv2d_mfma_kernel<WM, WN, MT, NT, C8_, R_, true, false> : conv2d_mfma_kernel<WM, WN, MT, NT, C8_, R_, false, false>; \
    }
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

} // namespace

namespace {
struct MfmaOverride {
    int bn = 0;     // 32 | 64 | 128, 0 = heuristic
    int splitK = 0; // >= 1, 0 = heuristic
};
} // namespace

static int make_conv2d_mfma_plan_ex(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, const MfmaOverride& ov,
                                    snnhip_plan** out) {
    // routing: GEMM-shaped layers only (north star: "MFMA used only for the dense 3x3/1x1 GEMM-shaped convs");
    // SNNHIP_CONV=generic|mfma forces a path (tests exercise both on the same inputs)
    const char* force = getenv("SNNHIP_CONV");
    if (force && strcmp(force, "generic") == 0) return SNNHIP_E_UNSUPPORTED;
    const bool forced = force && strcmp(force, "mfma") == 0;
    // measured (tools/bench_layers.py): even IC = 3 layers (one 8-channel chunk, 5/8 of it padding) run 1.4-2.3x faster here than
    // on the VALU kernel, so only the channel-thin outputs (OC < 16: most of a 32-wide MFMA column block would be padding) stay there
    // and the 16-wide thin layers (OC < 32 with IC < 32, e.g. ESPCN's 16->16: half of the narrowest 32-wide block is padding) measured
    // faster there too (241 vs 276 us at 1080p)
    const bool f16 = g.dtype == SNNHIP_F16; // fp16 tensors: this is the only general convolution kernel, it takes every shape
    if (!forced && !f16 && (g.OC < 16 || (g.OC < 32 && g.IC < 32))) return SNNHIP_E_UNSUPPORTED;
    const double inCount = static_cast<double>(g.N) * (g.preMode ? g.srcH : g.H) * (g.preMode ? g.srcW : g.W) * g.IC;
    const double outCount = static_cast<double>(g.N) * g.OH * g.OW * g.OC;
    if (inCount >= 2147483647.0 || outCount >= 2147483647.0) return SNNHIP_E_UNSUPPORTED; // 32-bit element offsets in the kernel

    const int taps = g.kh * g.kw;
    // channels per LDS chunk: 16 (8 when IC <= 8).  Wider chunks (the kernel is instantiated up to 64 channels; SNNHIP_CONV_C8=4|8 selects
    // them for pointwise stride-1 layers) were measured on the MobileNetV2 layers: fewer barriers per MFMA, but the 4x staging registers and
    // LDS cost more residency than they save (1x1 960->320 @7x7 b32: 58.7 -> 51.7 us, 144->24 @56x56: 27.9 -> 45.3 us), so 16 stays the default
    const int CH = f16 ? 8 : 4;                                  // channels per 16-byte slot
    int C8 = g.IC <= 2 * CH ? 1 : 2;
    // tap-pair mode (C8 = 0, see the kernel): channel-thin inputs with at least two taps; SNNHIP_CONV_PAIR=0 keeps the channel-chunk path
    const char* pairEnv = getenv("SNNHIP_CONV_PAIR");
    if (g.IC <= CH && taps >= 2 && !(pairEnv && atoi(pairEnv) == 0)) C8 = 0;
    if (const char* e = getenv("SNNHIP_CONV_C8"))
        if (taps == 1 && g.sh == 1 && g.sw == 1 && (atoi(e) == 4 || atoi(e) == 8) && g.IC >= 2 * CH * atoi(e)) C8 = atoi(e);
    int Qs = C8 ? 2 * C8 : 1;                                    // 16-byte slots per staged pixel
    int ICc = C8 ? 2 * CH * C8 : CH;                             // channels per LDS chunk: 16 fp32 / 32 fp16 (64 bytes per pixel either way)

    if (g.sw < 1 || g.sw > 2 || g.sh < 1 || g.sh > 2) return SNNHIP_E_UNSUPPORTED;

    // pixel tile: TB x TH x TW = 128, minimising padded pixels, then the staged halo
    struct TileLayout {
        int tileH, tileW, rowPitch, imgPitch, evenCols, total;
        size_t ldsBytes;
    };
    auto layout = [&](int TBs, int THs, int TWs) {
        const int TB = 1 << TBs, TH = 1 << THs, TW = 1 << TWs;
        TileLayout L;
        L.tileH = (TH - 1) * g.sh + g.kh;
        L.tileW = (TW - 1) * g.sw + g.kw;
        L.evenCols = g.sw == 2 ? (L.tileW + 1) / 2 : 0;
        L.rowPitch = L.tileW;
        if (TW < 32)  // lanes of one 32-lane half span several tile rows: their row step must be == TW (mod 16)
            while ((g.sh * L.rowPitch) % 16 != TW % 16) ++L.rowPitch;
        L.imgPitch = round_up(L.tileH * L.rowPitch, 16);
        L.total = TB * L.tileH * L.tileW * Qs;
        L.ldsBytes = static_cast<size_t>(2) * TB * L.imgPitch * Qs * 16;
        return L;
    };
    static const int shapes[][3] = {{0, 3, 4}, {0, 2, 5}, {0, 4, 3}, {1, 3, 3}, {2, 2, 3}, {3, 2, 2}, {0, 1, 6}, {0, 0, 7}};
    int best = -1;
    double bestCost = 0;
    size_t bestLds = 0;
    auto choose = [&]() {
        best = -1;
        const bool oneChunk = up_div(g.IC, ICc) == 1; // then the block stages once: no second buffer
        for (int s = 0; s < static_cast<int>(sizeof(shapes) / sizeof(shapes[0])); ++s) {
            const int TB = 1 << shapes[s][0], TH = 1 << shapes[s][1], TW = 1 << shapes[s][2];
            const TileLayout L = layout(shapes[s][0], shapes[s][1], shapes[s][2]);
            const size_t lds = oneChunk ? L.ldsBytes / 2 : L.ldsBytes;
            if (L.total > (C8 ? 9 : 5) * 256 || lds > 150 * 1024) continue; // staging registers / LDS
            const double tiles = static_cast<double>(up_div(g.N, TB)) * up_div(g.OH, TH) * up_div(g.OW, TW);
            double cost = tiles * (128.0 * taps + L.total / static_cast<double>(Qs) * 0.5); // MFMA work dominates, staging breaks ties
            if (lds > 80 * 1024) cost *= 1.4;                                 // only one block per CU would fit
            if (best < 0 || cost < bestCost) {
                best = s;
                bestCost = cost;
                bestLds = lds;
            }
        }
    };
    choose();
    // stride-2 halo tiles are ~4x the output tile: with 32-channel (fp32: 16) chunks they take > 80 KB and leave one block per CU (ResNet's
    // downsampling 3x3 convs ran at 28 TF/s fp32 / 90 TF/s fp16).  Half-width chunks double the barriers but keep 2-3 blocks resident.
    if (C8 == 2 && (best < 0 || bestLds > 80 * 1024) && !getenv("SNNHIP_CONV_WIDE_CHUNKS")) {
        C8 = 1;
        Qs = 2;
        ICc = 2 * CH;
        choose();
    }
    if (best < 0) return SNNHIP_E_UNSUPPORTED;
    MfmaParams p{};
    p.N = g.N; p.H = g.H; p.W = g.W; p.IC = g.IC; p.OC = g.OC; p.kh = g.kh; p.kw = g.kw; p.sh = g.sh; p.sw = g.sw;
    p.padx = g.padx; p.pady = g.pady; p.padMode = g.padMode; p.useBN = g.useBN; p.OH = g.OH; p.OW = g.OW;
    p.preMode = g.preMode; p.preX = g.preX; p.preY = g.preY; p.preShift = g.preMode ? g.preShift : 0;
    p.srcH = g.preMode ? g.srcH : g.H;
    p.srcW = g.preMode ? g.srcW : g.W;
    p.res = nullptr;
    p.ac2 = make_act_cfg(g.addAct >= 0 ? g.addAct : SNNHIP_ACT_NONE, g.addLeaky);
    p.TBs = shapes[best][0]; p.THs = shapes[best][1]; p.TWs = shapes[best][2];
    const int TB = 1 << p.TBs, TH = 1 << p.THs, TW = 1 << p.TWs;
    const TileLayout L = layout(p.TBs, p.THs, p.TWs);
    p.magicW = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(L.tileW) - 1) / static_cast<unsigned>(L.tileW));
    p.magicH = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(L.tileH) - 1) / static_cast<unsigned>(L.tileH));
    p.tileH = L.tileH; p.tileW = L.tileW; p.rowPitch = L.rowPitch; p.imgPitch = L.imgPitch; p.evenCols = L.evenCols;
    p.tilesX = up_div(g.OW, TW);
    p.tilesY = up_div(g.OH, TH);
    p.nChunks = up_div(g.IC, ICc);
    p.total = L.total;
    p.bufFloats = TB * L.imgPitch * Qs * 4;
    const int rNeed = up_div(p.total, 256);
    const int R = (rNeed <= 3 && C8 <= 2) ? 3 : ((rNeed <= 5 && C8 <= 4) ? 5 : 9);

    // block N: minimise (rounds of blocks over the CUs) x (work per block); narrower blocks re-stage A more often
    int BN = 128;
    {
        const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
        const double mtiles = static_cast<double>(p.tilesX) * p.tilesY * up_div(g.N, TB);
        double bestT = 0;
        const int cands[3] = {128, 64, 32};
        // fp16: the autotuner (SNNHIP_CONV_TUNE) picks the 64-wide kernel on most layers, even the deepest U-Net ones -- its 2x1 register block
        // leaves room for 3-4 resident blocks per CU where the 2x2 block of the 128-wide kernel (234 VGPRs) allows two
        const double penaltyF32[3] = {1.0, 1.08, 1.25}, penaltyF16[3] = {1.15, 1.0, 1.2};
        const double* penalty = f16 ? penaltyF16 : penaltyF32;
        for (int c = 0; c < 3; ++c) {
            const double blocks = mtiles * (round_up(g.OC, cands[c]) / cands[c]);
            const double t = std::ceil(blocks / cus) * cands[c] * penalty[c];
            if (c == 0 || t < bestT) {
                bestT = t;
                BN = cands[c];
            }
        }
    }
    if (const char* e = getenv("SNNHIP_CONV_BN")) // experiments: force the block's output-channel width
        if (atoi(e) == 32 || atoi(e) == 64 || atoi(e) == 128) BN = atoi(e);
    if (ov.bn) BN = ov.bn;
    p.OCp = round_up(g.OC, BN);
    // split-K: deep-K layers with few output tiles (ResNet 14x14 / 7x7 stages at batch 32, MobileNetV2's last pointwise convs) leave most
    // CUs with at most one wave per SIMD; splitting the channel chunks over blockIdx.z gives every SIMD 2+ waves.  Partial sums go to a
    // workspace and a second (element-wise, deterministic) pass applies bias/BN/activation.  SNNHIP_CONV_SPLITK=n forces n (1 = off).
    p.splitK = 1;
    {
        const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
        const double blocks = static_cast<double>(p.tilesX) * p.tilesY * up_div(g.N, TB) * (p.OCp / BN);
        int want = 1;
        if (blocks < 1.5 * cus && p.nChunks >= 8) want = static_cast<int>(std::ceil(2.0 * cus / blocks));
        if (want > 8) want = 8;
        if (want > p.nChunks / 4) want = p.nChunks / 4;
        if (const char* e = getenv("SNNHIP_CONV_SPLITK")) want = atoi(e);
        if (ov.splitK) want = ov.splitK;
        if (want < 1 || g.act == SNNHIP_ACT_SILU_QUIRK) want = 1; // the quirk couples 4 adjacent pixels in the epilogue
        if (want > p.nChunks) want = p.nChunks;
        p.chunksPerSplit = up_div(p.nChunks, want);
        p.splitK = up_div(p.nChunks, p.chunksPerSplit);
    }
    // fp16 output tile through LDS (see the kernel's epilogue): needs whole 8-channel vectors and the direct (non split-K) epilogue
    p.ldsEpi = (f16 && p.splitK == 1 && g.OC % 8 == 0 && !getenv("SNNHIP_CONV_DIRECT_STORE")) ? 1 : 0;
    size_t ldsNeed = p.chunksPerSplit == 1 ? L.ldsBytes / 2 : L.ldsBytes; // one chunk per block: no second staging buffer
    if (p.ldsEpi) ldsNeed = std::max(ldsNeed, static_cast<size_t>(128) * (BN + 8) * 2);
    const bool simple = act_is_simple(g.act);
    KernelFn fn = nullptr;
    if (BN == 128) fn = pick_kernel<2, 2, 2, 2>(C8, R, simple, f16, taps);
    if (BN == 64) fn = pick_kernel<2, 2, 2, 1>(C8, R, simple, f16, taps);
    if (BN == 32) fn = pick_kernel<4, 1, 1, 1>(C8, R, simple, f16, taps);
    if (!fn) return SNNHIP_E_UNSUPPORTED;

    auto* plan = new MfmaConvPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * taps);
    plan->epi4 = epi4;
    plan->p = p;
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->kernel = fn;
    plan->ldsBytes = ldsNeed;
    plan->fusedAdd = g.addAct >= 0;
    if (plan->fusedAdd) plan->numInputs = 2;
    plan->grid = dim3(p.tilesX * p.tilesY * up_div(g.N, TB), p.OCp / BN, p.splitK);
    if (p.splitK > 1) {
        void* ws = nullptr;
        const size_t wsBytes = static_cast<size_t>(p.splitK) * g.N * g.OH * g.OW * g.OC * sizeof(float);
        if (hipMalloc(&ws, wsBytes) != hipSuccess) {
            set_error("conv2d_mfma: split-K workspace of %zu bytes", wsBytes);
            delete plan;
            return SNNHIP_E_HIP;
        }
        plan->deviceAllocs.push_back(ws);
        plan->d_ws = static_cast<float*>(ws);
    }
    if (ldsNeed > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsNeed));
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(%zu) failed: %s", ldsNeed, hipGetErrorString(e));
            delete plan;
            return SNNHIP_E_HIP;
        }
    }

    // weights: Wp[chunk][tap][c8][h][OCp][j], ic = chunk*ICc + (c8*2 + h)*CH + j, j < CH (+ 10 zero steps: the prefetch ring reads up to 9
    // steps ahead); 16 bytes per (h, oc): 4 floats or 8 halfs (fp32 -> fp16 rounds to nearest; weights that went through the reference's
    // truncating convertToMediumPrecision are representable and convert exactly)
    const size_t steps = C8 ? static_cast<size_t>(p.nChunks) * taps * C8 : static_cast<size_t>((taps + 1) / 2);
    std::vector<float> wpk((steps + 10) * 2 * p.OCp * 4, 0.0f); // 16 bytes per (step, h, oc) in both precisions; 10 >= the deepest ring
    _Float16* wph = reinterpret_cast<_Float16*>(wpk.data());
    if (C8 == 0) { // tap-pair mode: Wp[step j][h][OCp][ic], tap = 2j + h
        for (int t = 0; t < taps; ++t)
            for (int j = 0; j < g.IC; ++j) {
                const size_t base = (static_cast<size_t>(t / 2) * 2 + (t & 1)) * p.OCp;
                for (int o = 0; o < g.OC; ++o) {
                    const float wv = w_oihw[(static_cast<size_t>(o) * g.IC + j) * taps + t];
                    if (f16) wph[(base + o) * 8 + j] = static_cast<_Float16>(wv);
                    else wpk[(base + o) * 4 + j] = wv;
                }
            }
    }
    for (int chunk = 0; chunk < p.nChunks; ++chunk)
        for (int t = 0; t < taps; ++t)
            for (int c8 = 0; c8 < C8; ++c8)
                for (int hh = 0; hh < 2; ++hh)
                    for (int j = 0; j < CH; ++j) {
                        const int ic = chunk * ICc + (c8 * 2 + hh) * CH + j;
                        if (ic >= g.IC) continue;
                        const size_t base = (((static_cast<size_t>(chunk) * taps + t) * C8 + c8) * 2 + hh) * p.OCp;
                        for (int o = 0; o < g.OC; ++o) {
                            const float wv = w_oihw[(static_cast<size_t>(o) * g.IC + ic) * taps + t];
                            if (f16) wph[(base + o) * 8 + j] = static_cast<_Float16>(wv);
                            else wpk[(base + o) * 4 + j] = wv;
                        }
                    }
    std::vector<float> epiP(static_cast<size_t>(p.OCp) * 4, 0.0f);
    std::memcpy(epiP.data(), epi4.data(), sizeof(float) * 4 * static_cast<size_t>(g.OC));
    int rc = plan->upload(wpk.data(), wpk.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epiP.data(), epiP.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = g.N; plan->inDims[1] = p.srcH; plan->inDims[2] = p.srcW; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->flops = 2.0 * taps * g.IC * g.OC * static_cast<double>(g.OH) * g.OW * g.N;
    plan->bytes = 4.0 * (static_cast<double>(g.N) * g.H * g.W * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC +
                         static_cast<double>(g.OC) * g.IC * taps);
    char buf[256];
    char chunkDesc[24];
    if (C8) snprintf(chunkDesc, sizeof(chunkDesc), "chunk=%d", ICc);
    else snprintf(chunkDesc, sizeof(chunkDesc), "tap-pairs");
    snprintf(buf, sizeof(buf), "conv2d_mfma_%s k=%dx%d s=%d ic=%d oc=%d tile=%dx%dx%dpx x %doc %s lds=%zuB splitK=%d",
             f16 ? "f16_32x32x16" : "f32_32x32x2", g.kh, g.kw, g.sh, g.IC, g.OC, TB, TH, TW, BN, chunkDesc, ldsNeed, p.splitK);
    plan->dtype = g.dtype;
    const double esz = f16 ? 2.0 : 4.0;
    plan->bytes = esz * (static_cast<double>(g.N) * p.srcH * p.srcW * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC + static_cast<double>(g.OC) * g.IC * taps);
    plan->desc = buf;
    if (g.preMode) plan->desc += " +pad(" + std::string(g.preMode == SNNHIP_PAD_REFLECT ? "reflect" : g.preMode == SNNHIP_PAD_REPLICATE ? "replicate" : "constant") + ")";
    if (g.preMode && g.preShift) plan->desc += " +upsample(x2)";
    if (plan->fusedAdd) {
        plan->desc += " +add";
        plan->bytes += esz * static_cast<double>(g.N) * g.OH * g.OW * g.OC; // the residual is read once
    }
    *out = plan;
    return SNNHIP_OK;
}

// Public entry: the heuristic configuration, or -- SNNHIP_CONV_TUNE=1 -- the fastest of a few (block width, split-K) candidates timed on the
// device at plan creation (5 launches each on scratch tensors; the winner per geometry is cached for the life of the process).  The
// heuristics were fitted on a handful of shapes; measured on the ResNet-18 body (fp16, batch 32) the best candidate is 10-25 % faster on
// the 28x28 / 14x14 / 7x7 stages, where block count, residency and the split-K reduce pass trade against each other.
int make_conv2d_mfma_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    const char* tune = getenv("SNNHIP_CONV_TUNE");
    if (!tune || atoi(tune) == 0 || getenv("SNNHIP_CONV_BN") || getenv("SNNHIP_CONV_SPLITK")) return make_conv2d_mfma_plan_ex(ctx, g, w_oihw, epi4, MfmaOverride(), out);
    char key[256];
    snprintf(key, sizeof(key), "%d.%d.%d.%d.%d.%d.%d.%d.%d.%d.%d.%d.%d.%d.%d.%d.%d", g.N, g.H, g.W, g.IC, g.OC, g.kh, g.kw, g.sh, g.sw, g.padx, g.pady, g.dtype,
             g.preMode, g.preShift, g.srcH, g.srcW, g.addAct >= 0);
    static std::map<std::string, MfmaOverride> cache;
    auto it = cache.find(key);
    if (it != cache.end()) return make_conv2d_mfma_plan_ex(ctx, g, w_oihw, epi4, it->second, out);

    const size_t esz = g.dtype == SNNHIP_F16 ? 2 : 4;
    const size_t inBytes = static_cast<size_t>(g.N) * (g.preMode ? g.srcH : g.H) * (g.preMode ? g.srcW : g.W) * g.IC * esz;
    const size_t outBytes = static_cast<size_t>(g.N) * g.OH * g.OW * g.OC * esz;
    void *dx = nullptr, *dy = nullptr, *dr = nullptr;
    if (hipMalloc(&dx, inBytes) != hipSuccess || hipMalloc(&dy, outBytes) != hipSuccess || (g.addAct >= 0 && hipMalloc(&dr, outBytes) != hipSuccess)) {
        if (dx) (void) hipFree(dx);
        if (dy) (void) hipFree(dy);
        return make_conv2d_mfma_plan_ex(ctx, g, w_oihw, epi4, MfmaOverride(), out);
    }
    (void) hipMemsetAsync(dx, 0, inBytes, ctx->stream);
    if (dr) (void) hipMemsetAsync(dr, 0, outBytes, ctx->stream);
    snnhip_tensor tx, ty, tr;
    tx.ctx = ty.ctx = tr.ctx = ctx;
    tx.n = g.N; tx.h = g.preMode ? g.srcH : g.H; tx.w = g.preMode ? g.srcW : g.W; tx.c = g.IC; tx.dtype = g.dtype; tx.data = static_cast<float*>(dx);
    ty.n = g.N; ty.h = g.OH; ty.w = g.OW; ty.c = g.OC; ty.dtype = g.dtype; ty.data = static_cast<float*>(dy);
    tr = ty;
    tr.data = static_cast<float*>(dr);
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    MfmaOverride best;
    float bestMs = -1.0f;
    const int bns[4] = {0, 32, 64, 128};
    const int splits[4] = {0, 1, 2, 4};
    std::string seen;
    for (int bi = 0; bi < 4; ++bi)
        for (int si = 0; si < 4; ++si) {
            if (bns[bi] > round_up(g.OC, 32) && bns[bi] != 32) continue; // wider than the layer: pure padding
            MfmaOverride ov;
            ov.bn = bns[bi];
            ov.splitK = splits[si];
            snnhip_plan* cand = nullptr;
            if (make_conv2d_mfma_plan_ex(ctx, g, w_oihw, epi4, ov, &cand) != SNNHIP_OK) continue;
            if (seen.find("|" + cand->desc + "|") != std::string::npos) { // same configuration as an earlier candidate
                delete cand;
                continue;
            }
            seen += "|" + cand->desc + "|";
            const snnhip_tensor* ins[2] = {&tx, &tr};
            const int nIn = g.addAct >= 0 ? 2 : 1;
            bool ok = cand->run(ins, nIn, &ty) == SNNHIP_OK; // warm-up
            float ms = 0.0f;
            for (int round = 0; round < 2 && ok; ++round) { // the better of two 5-launch timings: clocks wander on a busy box
                (void) hipEventRecord(e0, ctx->stream);
                for (int r = 0; r < 5 && ok; ++r) ok = cand->run(ins, nIn, &ty) == SNNHIP_OK;
                (void) hipEventRecord(e1, ctx->stream);
                float t = 0.0f;
                if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) ok = false;
                if (round == 0 || t < ms) ms = t;
            }
            delete cand;
            if (ok && (bestMs < 0.0f || ms < bestMs)) {
                bestMs = ms;
                best = ov;
            }
        }
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipFree(dx);
    (void) hipFree(dy);
    if (dr) (void) hipFree(dr);
    cache[key] = best;
    return make_conv2d_mfma_plan_ex(ctx, g, w_oihw, epi4, best, out);
}

} // namespace snnhip
