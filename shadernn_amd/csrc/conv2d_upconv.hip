// conv2d_upconv.hip -- fp16 "up-convolution" of the fast-neural-style networks (BASELINE configs[4]): nearest x2 UpSampling2D -> reflect Pad(1) ->
// Conv2D 3x3 stride 1 (64 -> 32 and 128 -> 64 channels in the zoo graph), evaluated on the LOW-RESOLUTION tensor.
//
// Round 2 ran these layers on conv2d_wide_f16 with the upsampling and the pad resolved in its staging addresses (rule D): 1103 + 870 us per 16
// images at 0.24 / 0.33 of the fp16 matrix peak, the two most expensive kernels of the graph.  That form multiplies every low-resolution pixel 9 x 4
// times.  But a 3x3 window over a x2-replicated image only ever sees a 2x2 block of DIFFERENT low-resolution pixels: for output pixel
// (2m + py, 2q + px) -- phase (py, px) of low-resolution position (m, q) -- the three rows of the window are rows {m-1, m, m} (py = 0) or
// {m, m, m+1} (py = 1) of the low-resolution tensor L, and the same for the columns.  So
//     out[2m + py][2q + px][oc] = sum_{a, b in {0,1}} sum_ic  Wp[py][px][a][b][oc][ic] * L[clamp(m - 1 + py + a)][clamp(q - 1 + px + b)][ic]
// with Wp the kernel taps that fall on the same low-resolution pixel ADDED UP (in fp32, then rounded to half once): 4 taps instead of 9 -- 2.25x
// fewer MFMAs and operand reads -- and the staged tile is the low-resolution halo tile (read once, not 4x replicated).  The reflect pad of 1 around
// the upsampled image is exactly a clamp of the low-resolution coordinate.  The reference's size rule keeps the PADDED extent as the output extent
// (2H + 2) with zeros beyond it: the last two output rows / columns see fewer taps.  They are the phases of ONE extra low-resolution row
// m = H (column q = W) with their own pre-summed weights (the taps that fall outside dropped) -- four weight classes (bulk / last row x bulk /
// last column), chosen per block.  Same operator contract otherwise (vk_upsampling2d_nearest.comp:43, padlayer.cpp:27-67,
// shadertemplate_vk_conv2d.comp:148-347: bias -> BN -> activation).
//
//   * block = one 32-column strip of the low-resolution grid of one image, marching down a segment of its rows 2 per iteration (= a 4 x 64 output
//     tile); LDS ring of 4 low-resolution rows (34 pixels x IC halfs, pixel pitch odd in 16-byte slots: conflict-free operand reads); the rows of
//     iteration it + 1 are requested while iteration it computes;
//   * wave = one phase (x one 32-channel tile): its 4 taps x IC/16 weight operands stay in registers for the whole strip (64 / 128 VGPRs), the 32
//     low-resolution pixels of a row are the MFMA's B operand; 8 IC/16 MFMAs per wave and iteration on two independent accumulators;
//   * the 4 x 64 x OC output tile leaves through LDS as 16-byte channel-contiguous vectors.
#include <cstring>
#include <type_traits>
#include <vector>

#include "epilogue.h"
#include "norm_fold.h"
#include "snnhip_internal.h"

namespace snnhip {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

struct UpParams {
    int N, srcH, srcW, IC, OC, OH, OW, useBN;
    int tilesX, segs, segRows; // strips of 32 low-resolution columns (+ 1 for column W), row segments per strip (+ 1 for row H), rows per segment (even)
    // chain rule F (Conv2D -> InstanceNorm): every block leaves ONE record {pixels, sum (v - bias), sum (v - bias)^2} per channel of the values it stored,
    // statRec[((n * gridDim.y + by) * blocksPerImage + block)][1 + 2 BN]; the last block of an image to finish adds the image's records up and writes the
    // norm's shift / mul (fold; the hand-off of norm_fold.h).  null = off
    float* statRec;
    NormFoldArgs fold;
    // graph rule I: the InstanceNorm in front, normAc(x * mul[n][c] + shift[n][c]), applied to the staged LOW-RESOLUTION values (once per pixel, not to the 4x
    // replicated ones); null = none.  Only the NORM instantiation reads these.
    const float* normShift;
    const float* normMul;
    ActCfg normAc;
};

#ifdef SNNHIP_UP_TRACE // experiment builds (tools/exp_one.sh)
#define UP_MARK(i) do { if (utrace) ustamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define UP_MARK(i) do { } while (0)
#endif
constexpr int kTP = 34; // staged pixels per low-resolution row: the strip's 32 + one on either side
#ifndef SNNHIP_UPCONV_OCC
#define SNNHIP_UPCONV_OCC 2 // waves per SIMD the 256-thread form is compiled for (3 = 168 VGPRs: the statistics accumulators then spill to scratch)
#endif

template <int ICS /* IC / 16: 4 | 8 */, int WNT /* 32-channel tiles per block: 1 | 2 */, bool NORM = false /* graph rule I (the 64-channel form: its thread keeps 8 shifts + 8 multipliers) */>
__global__ __launch_bounds__(256 * WNT, WNT == 1 ? SNNHIP_UPCONV_OCC : 2) void conv2d_upconv_kernel(UpParams p, ActCfg ac, const _Float16* __restrict__ x, const float4* __restrict__ wp,
                                                                                  const float4* __restrict__ epi, _Float16* __restrict__ y) {
    constexpr int Q = 2 * ICS, QP = Q + 1;   // 16-byte slots per pixel, and its (odd) pitch in LDS
    constexpr int MT = 2;                     // low-resolution rows per iteration
    constexpr int ROWF = kTP * QP * 4;        // floats per ring row
    constexpr int T = 256 * WNT;              // threads
    constexpr int BN = 32 * WNT, EP = BN + 8; // output channels per block; halfs per pixel of the output tile in LDS
    constexpr int EB = MT * kTP * Q;          // 16-byte elements per batch of MT rows
    constexpr int NR = (EB + T - 1) / T;      // staging rounds per batch (3)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* const otile = reinterpret_cast<_Float16*>(smem + 4 * ROWF); // [2 MT][64][EP] halfs
    float* const epiTab = smem + 4 * ROWF + (2 * MT * 64 * EP) / 2;       // [2][BN] scale, shift of this block's channels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int phase = wave & 3, py = phase >> 1, px = phase & 1, nt = wave >> 2;
    // block order: all marching blocks first (segment fastest within a strip), the one-row blocks of row H behind them.  Workgroups go to the eight
    // XCDs round-robin by index: with the one-row block as every (segs + 1)'th index and segs + 1 even, whole XCDs received nothing else (3
    // segments: XCDs 3 and 7 idle; round 5, odd segment counts measured 5-8 % slower than their even neighbours) -- and short jobs belong last
    const int bx = blockIdx.x;
    const int nseg = p.segs + 1;
    const int marching = p.N * p.tilesX * p.segs;
    const int strip = bx < marching ? bx / p.segs : bx - marching;
    const int sg = bx < marching ? bx - strip * p.segs : p.segs, tx = strip % p.tilesX, n = strip / p.tilesX;
    const bool lastRow = sg == p.segs, lastCol = tx == p.tilesX - 1;
    const int m0 = lastRow ? p.srcH : sg * p.segRows, mEnd = lastRow ? p.srcH + 1 : min(p.srcH, m0 + p.segRows);
    const int q0 = lastCol ? p.srcW : 32 * tx, qEnd = lastCol ? p.srcW + 1 : min(p.srcW, q0 + 32);
    const int nIter = (mEnd - m0 + MT - 1) / MT;
    const int ocb = blockIdx.y * BN;
    const int wclass = (lastRow ? 2 : 0) + (lastCol ? 1 : 0);

    // ---- this wave's weights (the MFMA's A operand): wq[a][b][cc] = 8 halfs {Wp[class][py][px][a][b][ocb + 32 nt + l32][16 cc + 8 h + j]}
    float4 wq[2][2][ICS];
    {
        const int ntg = blockIdx.y * WNT + nt, NT = p.OC / 32;
        const float4* wt = wp + ((static_cast<size_t>(wclass) * NT + ntg) * 4 + phase) * (4 * ICS * 64) + lane;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int cc = 0; cc < ICS; ++cc) wq[a][b][cc] = wt[((a * 2 + b) * ICS + cc) * 64];
    }
    // epilogue folded to one fma per value, act(acc * scale + shift) (conv2d_s2march.hip)
    if (tid < BN) {
        const float4 e4 = epi[ocb + tid];
        epiTab[tid] = p.useBN ? e4.y : 1.0f;
        epiTab[BN + tid] = p.useBN ? fmaf(e4.y, e4.x - e4.z, e4.w) : e4.x;
    }

    // ---- staging map of a batch of MT rows: element e = tid + T r -> row e / (34 Q), pixel, slot; the same for every batch
    const _Float16* xn = x + static_cast<size_t>(n) * p.srcH * p.srcW * p.IC;
    int colOfs[NR], ldsOfs[NR];
    unsigned rowBits = 0, liveBits = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = min(tid + T * r, EB - 1); // (elements past the batch re-read its last one and are not stored)
        const int row = e / (kTP * Q), rem = e - row * (kTP * Q);
        const int pxl = rem / Q, sl = rem - pxl * Q;
        const int sx = min(max(q0 - 1 + pxl, 0), p.srcW - 1); // the reflect pad of the upsampled image = a clamp down here
        colOfs[r] = sx * p.IC + 8 * sl;
        ldsOfs[r] = (row & 1) * ROWF + (pxl * QP + sl) * 4;
        rowBits |= static_cast<unsigned>(row & 1) << r;
        liveBits |= static_cast<unsigned>(tid + T * r < EB) << r;
    }
    static_assert(NR == 3, "three staging rounds per batch");
    float4 v0, v1, v2; // (named registers: as an array captured by the two lambdas below the rows in flight ended up in scratch memory)
    v0 = v1 = v2 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_one = [&](int r, int sy0, int sy1) {
        const int sy = ((rowBits >> r) & 1u) ? sy1 : sy0;
        return *reinterpret_cast<const float4*>(xn + static_cast<size_t>(sy) * p.srcW * p.IC + colOfs[r]);
    };
    auto load_batch = [&](int b) { // batch b = relative rows 2 b, 2 b + 1 (relative row r = low-resolution row m0 - 1 + r, clamped)
        const int sy0 = min(max(m0 - 1 + 2 * b, 0), p.srcH - 1), sy1 = min(max(m0 + 2 * b, 0), p.srcH - 1);
        v0 = load_one(0, sy0, sy1);
        v1 = load_one(1, sy0, sy1);
        v2 = load_one(2, sy0, sy1);
    };
    // graph rule I: half(act(x * mul + shift)) in fp32 on the 8 staged channels of an element -- the norm sweep's own arithmetic and rounding point.  T % Q == 0:
    // a thread's channel slot is the same for every element it ever stages, so its 8 shifts and 8 multipliers stay in registers; every staged pixel is a
    // real (clamped) pixel of the image, so every element is normalised; the activation kind is tested once per batch (conv2d_s2march.hip).
    static_assert(T % Q == 0, "a thread's staged elements share their channel slot");
    float nSh[8], nMu[8];
    const bool nRelu = NORM && p.normAc.act == SNNHIP_ACT_RELU;
    if constexpr (NORM) {
        const int slT = tid % Q;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            nSh[k] = p.normShift[n * p.IC + 8 * slT + k];
            nMu[k] = p.normMul[n * p.IC + 8 * slT + k];
        }
    }
    auto normalise = [&](float4& q, auto reluTag) {
        h8 hv = *reinterpret_cast<const h8*>(&q);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float f = fmaf(static_cast<float>(hv[k]), nMu[k], nSh[k]);
            hv[k] = static_cast<_Float16>(decltype(reluTag)::value ? fmaxf(f, 0.0f) : __builtin_amdgcn_fmed3f(fmaxf(f, f * p.normAc.alpha), p.normAc.lo, p.normAc.hi));
        }
        q = *reinterpret_cast<const float4*>(&hv);
    };
    auto store_batch = [&](int b) { // into ring rows 2 (b & 1), 2 (b & 1) + 1
        if constexpr (NORM) {
            if (nRelu) {
                normalise(v0, std::true_type{}); normalise(v1, std::true_type{}); normalise(v2, std::true_type{});
            } else {
                normalise(v0, std::false_type{}); normalise(v1, std::false_type{}); normalise(v2, std::false_type{});
            }
        }
        float* const dst = smem + (b & 1) * 2 * ROWF;
        if (liveBits & 1u) *reinterpret_cast<float4*>(dst + ldsOfs[0]) = v0;
        if (liveBits & 2u) *reinterpret_cast<float4*>(dst + ldsOfs[1]) = v1;
        if (liveBits & 4u) *reinterpret_cast<float4*>(dst + ldsOfs[2]) = v2;
    };

    // ---- MFMA B operand (low-resolution pixels): lane (l32, h), phase column px: pixel l32 + px + b, slot 2 cc + h of a ring row
    const int bofs = ((l32 + px) * QP + h) * 4;

    load_batch(0);
    store_batch(0);
    load_batch(1);
    store_batch(1);
    __syncthreads();

    const float* const et = epiTab + 32 * nt + 4 * h; // this lane's channel runs: 8 g + 4 h + k of the wave's 32-channel tile
    const bool actSimple = act_is_simple_dev(ac.act);
    const bool actNone = actSimple && ac.alpha == 1.0f && ac.lo == -__builtin_huge_valf() && ac.hi == __builtin_huge_valf();
    float shReg[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 sh = *reinterpret_cast<const float4*>(et + BN + 8 * g);
        shReg[4 * g] = sh.x; shReg[4 * g + 1] = sh.y; shReg[4 * g + 2] = sh.z; shReg[4 * g + 3] = sh.w;
    }
    // rule F: a thread's vectors of the store loop are pixels of ONE 8-channel column (T % (BN / 8) == 0): sums of (v - pivot) and (v - pivot)^2 of the
    // stored (rounded) values accumulate while they pass; pivot = the channel's bias (shift), the same in every block, so records simply add up
    float st1[8], st2[8], stN = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) st1[k] = st2[k] = 0.0f;
#ifdef SNNHIP_UP_TRACE
    const bool utrace = blockIdx.x == 900 && blockIdx.y == 0 && (tid == 0 || tid == 192);
    unsigned long long ustamp[8] = {};
#endif
    for (int it = 0; it < nIter; ++it) {
        const bool more = it + 1 < nIter;
        UP_MARK(0);
        if (more) load_batch(it + 2);
        UP_MARK(1);

        // ---- wave = phase (py, px) of low-resolution rows MT it, MT it + 1: row j takes taps a = 0, 1 from relative rows MT it + j + py + a
        f32x16 acc[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
        const int rb = __builtin_amdgcn_readfirstlane(MT * it + py);
        // steps s = (ri, b, cc): operand of relative row rb + ri, pixel column px + b, channel step cc; it feeds row j = ri with tap a = 0 and row
        // j = ri - 1 with tap a = 1.  The operand of step s + 1 is requested before the MFMAs of step s.
        constexpr int NS = 3 * 2 * ICS;
        float4 bop[2];
        auto read_step = [&](int s, float4& dst) {
            const int ri = s / (2 * ICS), b = (s / ICS) % 2, cc = s % ICS;
            dst = *reinterpret_cast<const float4*>(smem + ((rb + ri) & 3) * ROWF + bofs + (b * QP + 2 * cc) * 4);
        };
        read_step(0, bop[0]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) read_step(s + 1, bop[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int ri = s / (2 * ICS), b = (s / ICS) % 2, cc = s % ICS;
            if (ri < MT) acc[ri] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&wq[0][b][cc]), *reinterpret_cast<const h8*>(&bop[s & 1]), acc[ri], 0, 0, 0);
            if (ri >= 1) acc[ri - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&wq[1][b][cc]), *reinterpret_cast<const h8*>(&bop[s & 1]), acc[ri - 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue into the LDS tile: acc[j][4 g + k] = channel ocb + 32 nt + 8 g + 4 h + k of output pixel (2 j + py, 2 l32 + px) of the tile
        UP_MARK(2);
        // The lane's 16 shift values live in registers for the whole strip (shReg); only batch-norm layers read their scale per channel run from LDS.
        // (Read per run inside the loop -- four dependent LDS round trips -- the epilogue was 1 844 of an iteration's 4 600 cycles; holding all 32
        // table values of a BN layer next to the weights made the compiler park the prefetched rows in scratch memory.)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 sc = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            if (p.useBN) sc = *reinterpret_cast<const float4*>(et + 8 * g);
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
            float rv[MT][4];
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) rv[j][k] = fmaf(acc[j][4 * g + k], scv[k], shReg[4 * g + k]);
            if (actNone) { // (the style graphs: an InstanceNorm follows, the convolution itself has no activation -- three instructions a value saved
                           // in an epilogue that is 1 740 of an iteration's 4 970 cycles, phase trace of round 4)
            } else if (actSimple && ac.alpha == 1.0f) { // none / relu / relu6 with bounds: fmaxf(v, v * 1) is v
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) rv[j][k] = __builtin_amdgcn_fmed3f(rv[j][k], ac.lo, ac.hi);
            } else if (actSimple) { // (tested per channel run, not per value: a branch is a pipeline drain)
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) rv[j][k] = __builtin_amdgcn_fmed3f(fmaxf(rv[j][k], rv[j][k] * ac.alpha), ac.lo, ac.hi);
            } else {
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) rv[j][k] = epi_act(ac.act, ac.leaky, rv[j][k], 0.0f);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                h4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = static_cast<_Float16>(rv[j][k]);
                *reinterpret_cast<h4*>(otile + ((2 * j + py) * 64 + 2 * l32 + px) * EP + 32 * nt + 8 * g + 4 * h) = o;
            }
        }
        UP_MARK(3);
        __syncthreads(); // the tile is complete, and every wave is done with the ring rows that retire
        UP_MARK(4);

        // ---- the 4 x 64 tile leaves as 16-byte vectors, a pixel's BN channels contiguous
        const int mIt = m0 + MT * it;
        // the thread's 8 channels are the same for every vector it stores (T % (BN / 8) == 0): their pivots once per iteration, not once per vector -- where
        // the registers allow (the 128 -> 64 instantiation sits at 252 and re-reads them per vector)
        constexpr bool kHoistPivots = ICS <= 4;
        float pv[8];
        if (kHoistPivots && p.statRec) {
            const float4 pa = *reinterpret_cast<const float4*>(epiTab + BN + 8 * (tid % (BN / 8))), pb = *reinterpret_cast<const float4*>(epiTab + BN + 8 * (tid % (BN / 8)) + 4);
            pv[0] = pa.x; pv[1] = pa.y; pv[2] = pa.z; pv[3] = pa.w; pv[4] = pb.x; pv[5] = pb.y; pv[6] = pb.z; pv[7] = pb.w;
        }
#pragma unroll
        for (int q = 0; q < (2 * MT * 64 * (BN / 8)) / T; ++q) {
            const int vi = tid + T * q;
            const int pix = vi / (BN / 8), c8 = vi % (BN / 8);
            const int orow = pix >> 6, ocol = pix & 63;
            if (mIt + (orow >> 1) < mEnd && q0 + (ocol >> 1) < qEnd) {
                const float4 ov = *reinterpret_cast<const float4*>(otile + pix * EP + 8 * c8);
                *reinterpret_cast<float4*>(y + ((static_cast<size_t>(n) * p.OH + 2 * mIt + orow) * p.OW + 2 * q0 + ocol) * p.OC + ocb + 8 * c8) = ov;
                if (p.statRec) { // (uniform)
                    const h8 hv = *reinterpret_cast<const h8*>(&ov);
                    if (!kHoistPivots) {
                        const float4 pa = *reinterpret_cast<const float4*>(epiTab + BN + 8 * c8), pb = *reinterpret_cast<const float4*>(epiTab + BN + 8 * c8 + 4);
                        pv[0] = pa.x; pv[1] = pa.y; pv[2] = pa.z; pv[3] = pa.w; pv[4] = pb.x; pv[5] = pb.y; pv[6] = pb.z; pv[7] = pb.w;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float d = static_cast<float>(hv[k]) - pv[k];
                        st1[k] += d;
                        st2[k] = fmaf(d, d, st2[k]);
                    }
                    stN += 1.0f;
                }
            }
        }
        UP_MARK(5);
        if (more) store_batch(it + 2);
        UP_MARK(6);
        __syncthreads();
#ifdef SNNHIP_UP_TRACE
        if (utrace && it >= 4 && it < 7)
            printf("uptrace tid %d it %d: loads %llu mfma %llu epi %llu bar1 %llu stores %llu batch %llu bar2 %llu total %llu\n", tid, it, ustamp[1] - ustamp[0], ustamp[2] - ustamp[1],
                   ustamp[3] - ustamp[2], ustamp[4] - ustamp[3], ustamp[5] - ustamp[4], ustamp[6] - ustamp[5], __builtin_readcyclecounter() - ustamp[6], __builtin_readcyclecounter() - ustamp[0]);
#endif
    }

    if (!p.statRec) return; // (uniform)
    // ---- the block's record: the T / (BN / 8) threads of a channel column are added in a fixed order through LDS (ring and tile are dead)
    {
        constexpr int CPT = BN / 8, TPC = T / CPT; // channel columns, threads per column
        float* const red = smem;                   // [17][T]
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[k * T + tid] = st1[k];
            red[(8 + k) * T + tid] = st2[k];
        }
        red[16 * T + tid] = stN;
        __syncthreads();
        const int BPI = (p.segs + 1) * p.tilesX;
        float* const rec = p.statRec + (static_cast<size_t>(n * gridDim.y + blockIdx.y) * BPI + (tx * nseg + sg)) * (1 + 2 * BN);
        if (tid < BN) {
            const int col = tid >> 3, kk = tid & 7;
            float a1 = 0.0f, a2 = 0.0f, an = 0.0f;
            for (int j = 0; j < TPC; ++j) {
                a1 += red[kk * T + col + j * CPT];
                a2 += red[(8 + kk) * T + col + j * CPT];
                an += red[16 * T + col + j * CPT];
            }
            st_agent(rec + 1 + tid, a1);
            st_agent(rec + 1 + BN + tid, a2);
            if (tid == 0) st_agent(rec, an); // (every column of the block saw the same pixels)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the record is acknowledged before the block is counted (norm_fold.h)
        __syncthreads();
        if (tid == 0) {
            unsigned* cnt = p.fold.counter + n * gridDim.y + blockIdx.y;
            const unsigned prev = atomicAdd(cnt, 1u);
            const bool last = prev + 1u == static_cast<unsigned>(BPI);
            if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch (a replayed hipGraph)
            red[0] = last ? 1.0f : 0.0f;
        }
        __syncthreads();
        if (red[0] == 0.0f) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // one acquire per image, in its last block only
        // The fold is the tail of the launch (the last image's last block runs it when every other block is done), so it is built for latency:
        // thread = (channel, 1 of T / BN parts), four records (12 loads) in flight per thread -- one record per round trip to the coherence point
        // (the first version: a loop over ~110 records in BN threads) cost ~110 us per launch.  Fixed order: deterministic whichever block is last.
        {
            constexpr int PARTS = T / BN;
            const int ch = tid % BN, part = tid / BN;
            const float* r0 = p.statRec + static_cast<size_t>(n * gridDim.y + blockIdx.y) * BPI * (1 + 2 * BN);
            float a1 = 0.0f, a2 = 0.0f, an = 0.0f;
            for (int b0 = part; b0 < BPI; b0 += 4 * PARTS) {
                float t1[4], t2[4], tn[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int b = b0 + j * PARTS;
                    t1[j] = t2[j] = tn[j] = 0.0f;
                    if (b < BPI) {
                        const float* rb2 = r0 + static_cast<size_t>(b) * (1 + 2 * BN);
                        tn[j] = ld_agent(rb2);
                        t1[j] = ld_agent(rb2 + 1 + ch);
                        t2[j] = ld_agent(rb2 + 1 + BN + ch);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    an += tn[j];
                    a1 += t1[j];
                    a2 += t2[j];
                }
            }
            __syncthreads(); // (red[0] has been read by everyone)
            red[tid] = an;
            red[T + tid] = a1;
            red[2 * T + tid] = a2;
            __syncthreads();
            if (tid < BN) {
                an = a1 = a2 = 0.0f;
#pragma unroll
                for (int j = 0; j < PARTS; ++j) {
                    an += red[j * BN + tid];
                    a1 += red[T + j * BN + tid];
                    a2 += red[2 * T + j * BN + tid];
                }
                const float pivot = epiTab[BN + tid];
                const float dm = a1 / an, mean = pivot + dm;
                const float var = fmaxf(a2 / an - dm * dm, 0.0f);
                const float mu = p.fold.gamma[ocb + tid] / sqrtf(var + p.fold.eps);
                p.fold.mul[n * p.OC + ocb + tid] = mu;
                p.fold.shift[n * p.OC + ocb + tid] = p.fold.beta[ocb + tid] - mean * mu;
            }
        }
    }
}

struct UpconvPlan : ConvPlanBase {
    UpParams p;
    ActCfg ac;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    size_t ldsBytes = 0;
    dim3 grid, block;
    void (*kernel)(UpParams, ActCfg, const _Float16*, const float4*, const float4*, _Float16*) = nullptr;

    // chain rule F.  The records of this kernel are per BLOCK (a strip segment of any height, border strips of 2 pixels): not the regular tile grid
    // the norm's fold launches expect -- the statistics are only offered together with the in-kernel fold.
    bool enableTileStats() override {
        if (statPart) return true;
        if (snnhip::option("SNNHIP_NO_KERNEL_FOLD")) return false;
        const int BPI = (p.segs + 1) * p.tilesX, BN = static_cast<int>(p.OC / grid.y);
        void* buf = nullptr;
        if (snnhip::dev_malloc(&buf, static_cast<size_t>(p.N) * grid.y * BPI * (1 + 2 * BN) * sizeof(float)) != hipSuccess) return false;
        deviceAllocs.push_back(buf);
        statPart = p.statRec = static_cast<float*>(buf);
        statTilesX = BPI; statTilesY = 1; statTH = 0; statTW = 0;
        desc += " +tile-stats";
        return true;
    }
    bool tileStatsNeedKernelFold() const override { return true; }
    void disableTileStats() override {
        statPart = p.statRec = nullptr; // (the buffer stays with the plan's allocations)
        const size_t at = desc.rfind(" +tile-stats");
        if (at != std::string::npos) desc.erase(at);
    }
    bool enableNormFold(const NormFoldTarget& t) override {
        if (!statPart || p.fold.counter) return false;
        void* buf = nullptr;
        const size_t bytes = static_cast<size_t>(p.N) * grid.y * sizeof(unsigned);
        if (snnhip::dev_malloc(&buf, bytes) != hipSuccess) return false;
        deviceAllocs.push_back(buf);
        if (hipMemset(buf, 0, bytes) != hipSuccess) return false;
        p.fold.counter = static_cast<unsigned*>(buf);
        p.fold.gamma = t.gamma; p.fold.beta = t.beta; p.fold.shift = t.shift; p.fold.mul = t.mul; p.fold.eps = t.eps;
        desc += "+fold";
        return true;
    }

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "conv2d: expects 1 input, got %d", nIn);
        SNNHIP_REQUIRE(!p.statRec || p.fold.counter, "conv2d_upconv: block statistics were switched on without the in-kernel fold (no fold launch reads its records)");
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.srcH && x->w == p.srcW && x->c == p.IC, "conv2d: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h,
                       x->w, x->c, p.N, p.srcH, p.srcW, p.IC);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.OC, "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n,
                       out->h, out->w, out->c, p.N, p.OH, p.OW, p.OC);
        SNNHIP_LAUNCH(kernel, grid, block, ldsBytes, ctx->stream, p, ac, reinterpret_cast<const _Float16*>(x->data), reinterpret_cast<const float4*>(d_w),
                           reinterpret_cast<const float4*>(d_epi), reinterpret_cast<_Float16*>(out->data));
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

// kernel rows (columns) of the 3-tap window that fall on low-resolution row m - 1 + py + a, for the bulk and for the extra last row m = H (whose
// window reaches past the padded extent: those taps are dropped)
void tap_set(int py, int a, bool last, int out[3], int* n) {
    *n = 0;
    if (!last) {
        if (py == 0 && a == 0) out[(*n)++] = 0;
        if (py == 0 && a == 1) { out[(*n)++] = 1; out[(*n)++] = 2; }
        if (py == 1 && a == 0) { out[(*n)++] = 0; out[(*n)++] = 1; }
        if (py == 1 && a == 1) out[(*n)++] = 2;
    } else { // output rows 2H (py = 0: window rows 2H, 2H + 1 exist) and 2H + 1 (py = 1: only row 2H + 1)
        if (py == 0 && a == 0) out[(*n)++] = 0;
        if (py == 0 && a == 1) out[(*n)++] = 1;
        if (py == 1 && a == 0) out[(*n)++] = 0;
    }
}

} // namespace

// Tried by make_conv2d_mfma_plan (conv2d_mfma.hip) in front of conv2d_wide_f16; SNNHIP_E_UNSUPPORTED hands the layer on.
int make_conv2d_upconv_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    const char* force = snnhip::option("SNNHIP_CONV");
    const bool forced = force && strcmp(force, "upconv") == 0;
    // exactly: nearest x2 upsampling -> reflect pad 1 -> 3x3 stride 1, no padding of its own, output extent = the padded extent (size rule Q20)
    if (g.dtype != SNNHIP_F16 || g.kh != 3 || g.kw != 3 || g.sh != 1 || g.sw != 1 || g.preShift != 1 || g.preMode != SNNHIP_PAD_REFLECT || g.preX != 1 || g.preY != 1)
        return SNNHIP_E_UNSUPPORTED;
    if (g.padx != 0 || g.pady != 0 || g.H != 2 * g.srcH + 2 || g.W != 2 * g.srcW + 2 || g.OH != g.H || g.OW != g.W || g.addAct >= 0) return SNNHIP_E_UNSUPPORTED;
    // graph rule I: the 64-channel form only (the 128-channel one sits at 252 VGPRs), branch-free activations
    if (g.normShift && (g.IC != 64 || !act_is_simple(g.normAct) || snnhip::option("SNNHIP_NO_UPCONV_NORM"))) return SNNHIP_E_UNSUPPORTED;
    if ((g.IC != 64 && g.IC != 128) || g.srcH < 2 || g.srcW < 2 || g.act == SNNHIP_ACT_SILU_QUIRK) return SNNHIP_E_UNSUPPORTED;
    const int ICS = g.IC / 16, WNT = g.IC == 64 ? 1 : 2, BN = 32 * WNT;
    if (g.OC % BN != 0) return SNNHIP_E_UNSUPPORTED;
    if (static_cast<double>(g.N) * g.OH * g.OW * g.OC >= 2147483647.0 * 4 || static_cast<double>(g.N) * g.srcH * g.srcW * g.IC >= 2147483647.0) return SNNHIP_E_UNSUPPORTED;
    UpParams p = {};
    p.N = g.N; p.srcH = g.srcH; p.srcW = g.srcW; p.IC = g.IC; p.OC = g.OC; p.OH = g.OH; p.OW = g.OW; p.useBN = g.useBN;
    p.tilesX = up_div(g.srcW, 32) + 1;
    const int slots = std::max(1, ctx->props.multiProcessorCount) * (WNT == 1 ? SNNHIP_UPCONV_OCC : 1), strips = g.N * p.tilesX * (g.OC / BN);
    if (!forced && (g.srcH < 24 || static_cast<long>(strips) * up_div(g.srcH, 48) < ctx->props.multiProcessorCount)) return SNNHIP_E_UNSUPPORTED;
    {
        const char* fs = snnhip::option("SNNHIP_UPCONV_SEGS");
        int bestSegs = 1;
        double bestEff = -1.0;
        for (int s = 1; s <= 64; ++s) {
            const int rows = round_up(up_div(g.srcH, s), 2);
            const int segs = up_div(g.srcH, rows);
            if (segs != s) continue;
            if (s > 1 && rows < 16 && !fs) break;
            // blocks are handed out dynamically and do not finish together: what a launch loses is about half a block time at its end, not the
            // unfilled part of a last "round" (round 5, tools/bench_upconv.py --segs at Candy's micro-batch 16: the round model picked 3 segments
            // for both up-convolutions, 973 / 642 us; 6 segments 885 / 599 us, flat from 4 to 12).  The segs + 1'th block of a strip is one row.
            const double perSlot = static_cast<double>(strips) * segs / slots;
            const double eff = perSlot / (perSlot + 0.5) * rows / (rows + 4);
            if ((fs && atoi(fs) == s) || (!fs && eff > bestEff + 1e-9)) {
                bestEff = eff;
                bestSegs = segs;
                if (fs) break;
            }
        }
        p.segRows = round_up(up_div(g.srcH, bestSegs), 2);
        p.segs = up_div(g.srcH, p.segRows);
    }
    const int QP = 2 * ICS + 1;
    const size_t lds = static_cast<size_t>(4) * kTP * QP * 16 + static_cast<size_t>(4) * 64 * (BN + 8) * 2 + 2 * static_cast<size_t>(BN) * 4;
    p.normShift = g.normShift; p.normMul = g.normMul;
    p.normAc = make_act_cfg(g.normShift ? g.normAct : SNNHIP_ACT_NONE, g.normLeaky);
    auto fn = g.IC == 64 ? (g.normShift ? conv2d_upconv_kernel<4, 1, true> : conv2d_upconv_kernel<4, 1, false>) : conv2d_upconv_kernel<8, 2, false>;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
        set_error("conv2d_upconv: hipFuncSetAttribute(%zu) failed", lds);
        return SNNHIP_E_HIP;
    }
    auto* plan = new UpconvPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * 9);
    plan->epi4 = epi4;
    plan->p = p;
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->kernel = fn;
    plan->ldsBytes = lds;
    plan->grid = dim3(static_cast<unsigned>(p.tilesX) * (p.segs + 1) * g.N, static_cast<unsigned>(g.OC / BN));
    plan->block = dim3(256 * WNT);
    plan->dtype = SNNHIP_F16;
    // weights: Wp[class][32-channel tile][phase][tap a * 2 + b][cc][lane = 32 hh + m] x 8 halfs: the kernel taps that fall on low-resolution pixel (a, b)
    // of the phase's 2x2 block, summed in fp32 and rounded to half once
    const int NT = g.OC / 32;
    std::vector<float> wpk(static_cast<size_t>(4) * NT * 4 * 4 * ICS * 64 * 4, 0.0f);
    _Float16* wph = reinterpret_cast<_Float16*>(wpk.data());
    for (int cls = 0; cls < 4; ++cls)
        for (int ph = 0; ph < 4; ++ph)
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    int fys[3], fxs[3], ny = 0, nx = 0;
                    tap_set(ph >> 1, a, (cls & 2) != 0, fys, &ny);
                    tap_set(ph & 1, b, (cls & 1) != 0, fxs, &nx);
                    for (int oc = 0; oc < g.OC; ++oc)
                        for (int ic = 0; ic < g.IC; ++ic) {
                            float sum = 0.0f;
                            for (int i = 0; i < ny; ++i)
                                for (int j = 0; j < nx; ++j) sum += static_cast<float>(static_cast<_Float16>(w_oihw[((static_cast<size_t>(oc) * g.IC + ic) * 3 + fys[i]) * 3 + fxs[j]]));
                            const int ntile = oc / 32, m = oc % 32, cc = ic / 16, hh = (ic % 16) / 8, jj = ic % 8;
                            wph[((((static_cast<size_t>(cls) * NT + ntile) * 4 + ph) * 4 + (a * 2 + b)) * ICS + cc) * 64 * 8 + (hh * 32 + m) * 8 + jj] = static_cast<_Float16>(sum);
                        }
                }
    int rc = plan->upload(wpk.data(), wpk.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epi4.data(), epi4.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = g.N; plan->inDims[1] = g.srcH; plan->inDims[2] = g.srcW; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->flops = 2.0 * 9 * g.IC * g.OC * static_cast<double>(g.OH) * g.OW * g.N; // the ALGORITHMIC count of the 3x3 layer; executed: 4 / 9 of it
    plan->bytes = 2.0 * (static_cast<double>(g.N) * g.srcH * g.srcW * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC + static_cast<double>(g.OC) * g.IC * 9);
    char buf[360];
    snprintf(buf, sizeof(buf), "conv2d_mfma_upconv_f16_32x32x16 k=3x3 s=1 ic=%d oc=%d as 4 phases x 2x2 taps on the low-resolution tensor (pre-summed weights), row-marching strips=2x32 low-res px "
                               "(4x64 out) x %doc segments=%d x %d rows lds=%zuB mfma_flops=%.6g +pad(reflect) +upsample(x2)",
             g.IC, g.OC, BN, p.segs, p.segRows, lds, plan->flops * 4.0 / 9.0);
    plan->desc = buf;
    if (g.normShift) plan->desc = "instancenorm(act=" + std::to_string(g.normAct) + ", in the staging) -> " + plan->desc;
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
