// conv2d_rowmarch.hip -- the row-MARCHING form of conv2d_rowfold.hip (fp16 k x k stride-1 convolution with k * OC <= 32 and OC <= 4: the
// image-producing 9x9 32 -> 3 layer of the fast-neural-style networks, BASELINE configs[4]).  Same arithmetic idea -- the kernel's COLUMNS are
// folded into the GEMM's N (P[y][x'][fx][oc] = sum_{fy, ic} W * in, then out[y][x][oc] = sum_fx P[y][x + fx][fx][oc]) -- but built around what
// round 2's counters said about the tile kernel (profiles/r02_c5_*: FETCH 2.76 GB for a 1.17 GB input, WRITE 1.14 GB for a 0.11 GB output,
// 861 us at 4.4 TB/s of traffic it should not generate):
//
//   * a block owns a 64-column strip of ONE image and MARCHES down a segment of its rows, 8 output rows per iteration.  The 16 input rows an
//     iteration needs live in an LDS ring (two groups of 8 rows); the next iteration re-uses the younger group and only the 8 NEW rows are
//     fetched -- while the current iteration computes, into registers, and are written over the group that has just retired.  An input row is
//     read 64/56 = 1.14x (column halo) x (segment rows + 8) / segment rows (row halo) instead of 16/8 x 64/56 = 2.29x.
//   * the MFMA runs with the WEIGHTS as the A operand and the 32 pixels of a row tile as B, so a lane ends up with ITS pixel's 16 of the 32
//     (fx, oc) columns: the shift-add over fx becomes 16 + 8 ds_bpermute per row tile (lane x pulls column (fx, oc) from lane x + fx; the pull
//     that crosses into the next row tile comes from that tile's accumulator) + one exchange between the two lane halves.  No P tile in LDS, no
//     barrier between the MFMAs and the output: the wave is autonomous from its accumulators to its stores.
//   * a finished output row (56 pixels x OC halfs, contiguous in NHWC) is packed in a wave-private LDS line and leaves as 4-byte stores of
//     consecutive lanes (whole 64-byte segments) instead of three 2-byte stores per pixel.
// Operator contract as conv2d_rowfold.hip (shadertemplate_vk_conv2d.comp:148-347: padding modes, bias -> BN -> activation; the Pad layer in
// front fused into the staging, rule D; the InstanceNorm in front applied to the staged values, rule I).
#include <cstring>
#include <vector>

#include "epilogue.h"
#include "snnhip_internal.h"

namespace snnhip {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct RowmarchParams {
    int N, H, W, IC, OC, OH, OW, padx, pady, padMode, useBN;
    int preMode, preX, preY, srcH, srcW, preShift; // fused Pad / nearest x2 upsampling in front (ConvGeom)
    int tilesX, segs, segRows;                     // column strips, row segments per strip, output rows per segment (multiple of 8)
    const float* normShift;                        // InstanceNorm in front (graph rule I); null = none
    const float* normMul;
    ActCfg normAc;
};

constexpr int kTH = 8, kCols = 64;
#ifdef SNNHIP_RM_TRACE // experiment builds (tools/exp_one.sh): one block prints the s_memtime stamps of its phases
#define RM_MARK(i) do { if (rtrace) rstamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define RM_MARK(i) do { } while (0)
#endif

// Orders a wave's LDS writes before its later LDS reads of the same (wave-private) bytes through other lanes and pointer types.  NOT a fence: the
// release / acquire fence pair this started as compiles to s_waitcnt vmcnt(0) -- every output row then waited for the row prefetch issued at the top
// of the iteration AND for the previous row's stores to be acknowledged (phase trace: 10 300 of an iteration's 17 000 cycles in the epilogue).  A wave's
// LDS operations complete in order; what is needed is that the compiler keeps the order (the asm's memory clobber) and that the write has left the
// queue (lgkmcnt).
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

template <int K, int ICS /* IC / 16 */, int OC>
__global__ __launch_bounds__(256, 2) void conv2d_rowfold_march_kernel(RowmarchParams p, ActCfg ac, const _Float16* __restrict__ x, const float4* __restrict__ wp,
                                                                     const float4* __restrict__ epi, _Float16* __restrict__ y) {
    constexpr int TW = kCols - K + 1;          // output columns per strip
    constexpr int Q = 2 * ICS;                 // 16-byte slots per pixel
    constexpr int NK = K * ICS;                // K steps (weights kept in registers)
    constexpr int CPR = kCols * Q;             // 16-byte elements per staged row (256 or 128)
    constexpr int RPI = 256 / CPR;             // staged rows per round of 256 elements (1 or 2)
    constexpr int NRND = kTH / RPI;            // rounds per group of 8 rows (8 or 4)
    constexpr int ROWF = kCols * Q * 4;        // floats per ring row
    constexpr int SWS = Q == 4 ? 2 : 3;        // swizzle: slot ^= (column >> SWS) & (Q - 1)
    static_assert(K * OC <= 32 && OC >= 1 && OC <= 4, "columns (fx, oc) must fit one 32-wide MFMA tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* const scratch = reinterpret_cast<_Float16*>(smem + 16 * ROWF); // [4 waves][2 rows][64 px * 4 halfs]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int mt = blockIdx.x;
    const int seg = mt % p.segs, tx = (mt / p.segs) % p.tilesX, n = mt / (p.segs * p.tilesX);
    const int ox0 = tx * TW, oyS = seg * p.segRows, oyE = min(p.OH, oyS + p.segRows);
    const int ix0 = ox0 - p.padx, iyS = oyS - p.pady;
    const int nIter = (oyE - oyS + kTH - 1) / kTH;

    // ---- this lane's weights (the MFMA's A operand): step s = fy * ICS + c -> 8 halfs {W[oc][16c + 8h + j][fy][fx]} of row (fx, oc) = fx * OC + oc = l32
    float4 b[NK];
#pragma unroll
    for (int s = 0; s < NK; ++s) b[s] = wp[s * 64 + lane];

    // ---- staging of a group of 8 input rows (relative rows 8g .. 8g + 7 of the segment): thread -> column (tid / Q) % 64, slot tid % Q, row r * RPI + rsub
    // of each round r; column resolved once per thread, rows on the scalar unit (see conv2d_rowfold.hip)
    const _Float16* xn = x + static_cast<size_t>(n) * p.srcH * p.srcW * p.IC;
    const int sl = tid % Q, c = (tid / Q) % kCols;
    const int rsub = __builtin_amdgcn_readfirstlane(tid / CPR);
    int sx = resolve_nobranch(ix0 + c, p.W, p.padMode);
    if (p.preMode) {
        const int px = resolve_nobranch(sx - p.preX, p.srcW << p.preShift, p.preMode);
        sx = sx < 0 ? -1 : (px < 0 ? -1 : px >> p.preShift);
    }
    const bool colOk = sx >= 0;
    const int colOfs = (colOk ? sx : 0) * p.IC + 8 * sl;
    float* const ldsCol = smem + (c * Q + (sl ^ ((c >> SWS) & (Q - 1)))) * 4; // + ring row * ROWF
    float nShift[8], nMul[8];
    if (p.normShift) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            nShift[k] = p.normShift[static_cast<size_t>(n) * p.IC + 8 * sl + k];
            nMul[k] = p.normMul[static_cast<size_t>(n) * p.IC + 8 * sl + k];
        }
    }
    const bool nRelu = p.normAc.act == SNNHIP_ACT_RELU;

    float4 v[NRND];
    unsigned rowOkMask = 0;
    // The source row of a staged row is wave-uniform, but resolving it on the scalar unit (padding mode, fused Pad, fused UpSampling: ~60 SALU
    // instructions and two branches per row) made the row loop 1000 SALU instructions + 16 pipeline drains per iteration next to the wave's 72 MFMAs.
    // Lane r resolves row r of the group ONCE on the vector unit; a round reads its row back with v_readlane.
    auto load_group = [&](int g) { // issue the loads of group g (rows outside the padded image / the source read row 0 and are zeroed when stored)
        int syv = resolve_nobranch(iyS + g * kTH + (lane & 7), p.H, p.padMode);
        {
            const int py = resolve_nobranch(syv - p.preY, p.srcH << p.preShift, p.preMode);
            const int pre = syv < 0 ? -1 : (py < 0 ? -1 : py >> p.preShift);
            syv = p.preMode ? pre : syv;
        }
        rowOkMask = 0;
#pragma unroll
        for (int r = 0; r < NRND; ++r) {
            const int sy = __builtin_amdgcn_readlane(syv, r * RPI + rsub); // (r * RPI + rsub is wave-uniform)
            rowOkMask |= static_cast<unsigned>(sy >= 0) << r;
            v[r] = *reinterpret_cast<const float4*>(xn + static_cast<size_t>(max(sy, 0)) * p.srcW * p.IC + colOfs); // (a column outside reads column 0: masked when stored)
        }
    };
    auto store_group = [&](int g) { // normalise (rule I) and write the group into ring rows 8 (g & 1) ..
        if (p.normShift) {
            if (nRelu) {
#pragma unroll
                for (int r = 0; r < NRND; ++r) {
                    h8 hv = *reinterpret_cast<const h8*>(&v[r]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) hv[k] = static_cast<_Float16>(fmaxf(fmaf(static_cast<float>(hv[k]), nMul[k], nShift[k]), 0.0f));
                    v[r] = *reinterpret_cast<const float4*>(&hv);
                }
            } else {
#pragma unroll
                for (int r = 0; r < NRND; ++r) {
                    h8 hv = *reinterpret_cast<const h8*>(&v[r]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float f = fmaf(static_cast<float>(hv[k]), nMul[k], nShift[k]);
                        hv[k] = static_cast<_Float16>(__builtin_amdgcn_fmed3f(fmaxf(f, f * p.normAc.alpha), p.normAc.lo, p.normAc.hi));
                    }
                    v[r] = *reinterpret_cast<const float4*>(&hv);
                }
            }
        }
        float* const dst = ldsCol + ((g & 1) * kTH + rsub) * ROWF;
#pragma unroll
        for (int r = 0; r < NRND; ++r) {
            const bool live = ((rowOkMask >> r) & 1u) && colOk; // padding stays zero (the norm is not applied to it)
            const float4 o = make_float4(live ? v[r].x : 0.f, live ? v[r].y : 0.f, live ? v[r].z : 0.f, live ? v[r].w : 0.f);
            *reinterpret_cast<float4*>(dst + r * RPI * ROWF) = o;
        }
    };

    // ---- MFMA operand addressing (B operand = pixels): lane (l32, h) reads column t * 32 + l32, slot 2 c + h of a ring row
    int bofs[2][ICS];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int cc = 0; cc < ICS; ++cc) {
            const int col = t * 32 + l32, s = 2 * cc + h;
            bofs[t][cc] = (col * Q + (s ^ ((col >> SWS) & (Q - 1)))) * 4;
        }

    // ---- shift-add tables.  D layout of the 32x32 MFMA with A = weights: lane (l32, h) holds, for pixel l32, rows (fx, oc) index
    // n_i = 8 (i / 4) + 4 h + i % 4 in register i.  For register i this lane pulls from the lane of pixel l32 + fx(n_i) (same half): the value is
    // P[x + fx][fx][oc], its term of out[x][oc(n_i)].  fx / oc of a register differ between the two lane halves only.
#ifdef SNNHIP_RM_PULLS // (the form of rounds 2-5; round 6's epilogue below needs no tables)
    int pa[16];            // ds_bpermute byte address
    unsigned crossMask = 0; // bit i: the pulled pixel sits in the NEXT 32-column tile
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n0 = 8 * (i >> 2) + (i & 3), n1 = n0 + 4;
        const int fx = h ? n1 / OC : n0 / OC;
        const int src = l32 + fx;
        pa[i] = 4 * ((src & 31) + 32 * h);
        crossMask |= static_cast<unsigned>(src >= 32) << i;
    }
    const bool h1 = h != 0;
#endif
    // (uniform) none / relu / relu6 / leakyRelu go through the branch-free med3 form; the run-time switch of epi_act (tanh, sigmoid, ...) pulled
    // ~700 VALU instructions and ~200 branches into the iteration loop
    const bool actSimple = act_is_simple_dev(ac.act);

    // prologue: groups 0 and 1
    load_group(0);
    store_group(0);
    load_group(1);
    store_group(1);
    __syncthreads();

    // the epilogue rows {bias, bnScale, bnMean, bnBeta} of the OC channels, read once (wave-uniform: they live in SGPRs); indexed inside the loop they
    // were re-loaded every iteration, twelve scalar loads each waited out
    float4 eR[OC];
#pragma unroll
    for (int k = 0; k < OC; ++k) eR[k] = epi[k];
    const size_t rowHalfs = static_cast<size_t>(p.OW) * OC;
    const int validCols = min(TW, p.OW - ox0);
#ifdef SNNHIP_RM_TRACE
    const bool rtrace = blockIdx.x == 700 && (tid == 0 || tid == 192);
    unsigned long long rstamp[6] = {}, estamp[8] = {};
#endif
    for (int it = 0; it < nIter; ++it) {
        const bool more = it + 1 < nIter;
        RM_MARK(0);
        if (more) load_group(it + 2); // consumed after this iteration's MFMAs and epilogue
        RM_MARK(1);

        // ---- wave = output rows 2w, 2w + 1 of the iteration x both 32-column tiles
        f32x16 acc[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[j][t][i] = 0.0f;
        const int rbase = __builtin_amdgcn_readfirstlane(kTH * (it & 1) + 2 * wave);
        // operand reads one input row ahead of the MFMAs that consume them (left to the scheduler they sat right in front of their MFMAs behind an
        // lgkmcnt(0): sixteen exposed LDS round trips per iteration)
        float4 a[2][ICS][2];
        auto read_row = [&](int rr, float4 (&dst)[ICS][2]) {
            const float* rowp = smem + ((rbase + rr) & 15) * ROWF;
#pragma unroll
            for (int cc = 0; cc < ICS; ++cc)
#pragma unroll
                for (int t = 0; t < 2; ++t) dst[cc][t] = *reinterpret_cast<const float4*>(rowp + bofs[t][cc]);
        };
        read_row(0, a[0]);
#pragma unroll
        for (int rr = 0; rr < K + 1; ++rr) { // relative input row 8 it + 2 w + rr feeds output row j at tap fy = rr - j
            if (rr < K) read_row(rr + 1, a[(rr + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cc = 0; cc < ICS; ++cc) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int fy = rr - j;
                    if (fy < 0 || fy >= K) continue; // compile-time after unrolling
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&b[fy * ICS + cc]), *reinterpret_cast<const h8*>(&a[rr & 1][cc][t]), acc[j][t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        RM_MARK(2);
        __syncthreads(); // every wave is done with the older group: its ring rows may be overwritten (below, after the epilogue)
        RM_MARK(3);

        // ---- epilogue, per wave: shift-add by lane pulls, bias -> BN -> activation, row packed in the wave's LDS line, 4-byte stores
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = oyS + it * kTH + 2 * wave + j;
            if (oy >= oyE) continue; // (uniform)
            _Float16* const line = scratch + (wave * 2 + j) * (kCols * 4);
#ifndef SNNHIP_RM_PULLS // (round 6; -DSNNHIP_RM_PULLS builds the ds_bpermute form of rounds 2-5 for A/B runs)
            {
                // The shift-add over fx as a Horner walk over the 64 pixels of the strip at once.  Register i of a tile holds (fx, oc) column n0 = 8 (i / 4) + i % 4
                // in lane half 0 and n0 + 4 in half 1, for pixel l32 of ITS tile.  ONE v_permlane32_swap of the two tiles' register i (swap the upper half of
                // the first with the lower half of the second) leaves  X_i = column n0 of pixel `lane` (0..63)  and  Y_i = column n0 + 4 of pixel `lane`:
                // every (fx, oc) column as a 64-lane vector in pixel order.  Then  out[x][oc] = sum_fx P[x + fx][fx][oc]  is
                //     T = P[.][K-1][oc];  T = shl1(T) + P[.][fx][oc]  for fx = K-2 .. 0          (shl1: lane x takes lane x + 1 = DPP wave_shl:1, one VALU slot)
                // -- K - 1 shifted adds per channel for BOTH tiles, no ds_bpermute, no exchange between the lane halves, no pull across the tile boundary (the strip
                // is exactly the wave: lanes beyond TW - 1 compute sums that reach past it and are never stored).  Rounds 2-5: 45 ds_bpermute + 3 cross-half
                // shuffles per output row (phase trace: pulls + adds 630, exchange 335 of a row tile's ~1 800 cycles).
#ifdef SNNHIP_RM_TRACE
                if (rtrace && j == 0) estamp[0] = __builtin_readcyclecounter();
#endif
                float X[16], Y[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (8 * (i >> 2) + (i & 3) >= K * OC) continue; // (compile-time) neither half holds a real column in this register
                    const float lo = acc[j][0][i], hi = acc[j][1][i];
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
                    X[i] = __uint_as_float(sw[0]);
                    Y[i] = __uint_as_float(sw[1]);
                }
#ifdef SNNHIP_RM_TRACE
                if (rtrace && j == 0) estamp[1] = __builtin_readcyclecounter();
#endif
                float tot[OC];
#pragma unroll
                for (int k = 0; k < OC; ++k) {
                    float T = 0.0f;
#pragma unroll
                    for (int fx = K - 1; fx >= 0; --fx) {
                        const int m = fx * OC + k, i = 4 * (m >> 3) + (m & 3); // (compile-time) column m sits in register i, lane half (m >> 2) & 1
                        const float col = ((m >> 2) & 1) ? Y[i] : X[i];
                        if (fx == K - 1) T = col;
                        else T = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(T), 0x130 /* wave_shl:1 */, 0xf, 0xf, false)) + col;
                    }
                    tot[k] = T;
                }
#ifdef SNNHIP_RM_TRACE
                if (rtrace && j == 0) estamp[2] = __builtin_readcyclecounter();
#endif
#pragma unroll
                for (int k = 0; k < OC; ++k) tot[k] = epi_affine(tot[k], eR[k], p.useBN);
                if (actSimple) { // (tested once per row, not per value: a branch is a pipeline drain)
#pragma unroll
                    for (int k = 0; k < OC; ++k) tot[k] = __builtin_amdgcn_fmed3f(fmaxf(tot[k], tot[k] * ac.alpha), ac.lo, ac.hi);
                } else {
#pragma unroll
                    for (int k = 0; k < OC; ++k) tot[k] = epi_act(ac.act, ac.leaky, tot[k], 0.0f);
                }
#pragma unroll
                for (int k = 0; k < OC; ++k) line[lane * OC + k] = static_cast<_Float16>(tot[k]);
            }
#else
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // Register i holds column n0 = 8 (i / 4) + i % 4 in lane half 0 and n0 + 4 in half 1: channel n0 % OC resp. (n0 + 4) % OC.  Both halves add
                // their pull to accumulator n0 % OC -- ONE add, no select: in half 1 accumulator r therefore stands for channel (r + 4) % OC, undone
                // when the halves are combined.
                float o[OC];
#pragma unroll
                for (int k = 0; k < OC; ++k) o[k] = 0.0f;
                // all pulls of the row tile are issued before the first is used (one at a time each waited out its LDS round trip: ~50 exposed waits
                // per iteration)
#ifdef SNNHIP_RM_TRACE
                if (rtrace && j == 0 && t == 0) estamp[0] = __builtin_readcyclecounter();
#endif
                float pv[16], pn[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (8 * (i >> 2) + (i & 3) >= K * OC) continue; // (compile-time) neither half holds a real column in this register
                    // (element copies first: __builtin_bit_cast applied to the vector-element expression itself pulled element 0 whatever i was)
                    const float own = acc[j][t][i];
                    pv[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(pa[i], __float_as_int(own)));
                    if (t == 0) {
                        const float nxt = acc[j][1][i];
                        pn[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(pa[i], __float_as_int(nxt)));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int n0 = 8 * (i >> 2) + (i & 3), n1 = n0 + 4;
                    if (n0 >= K * OC) continue;
                    float q = pv[i];
                    if (t == 0) q = ((crossMask >> i) & 1u) ? pn[i] : q;
                    // half 1's column n1 may be one of the zero-weight padding columns: its pull reaches past the receptive field (fx >= K), where
                    // 0 * inf would be NaN -- dropped
                    if (n1 >= K * OC) q = h1 ? 0.0f : q;
                    o[n0 % OC] += q;
                }
#ifdef SNNHIP_RM_TRACE
                if (rtrace && j == 0 && t == 0) estamp[1] = __builtin_readcyclecounter();
#endif
                float tot[OC]; // (the exchange runs with every lane active: a pull from a disabled lane returns 0)
#pragma unroll
                for (int k = 0; k < OC; ++k) tot[k] = o[k] + __shfl_xor(o[((k - 4) % OC + OC) % OC], 32); // half 0: channel k of the other half sits in ITS accumulator (k - 4) mod OC
#ifdef SNNHIP_RM_TRACE
                if (rtrace && j == 0 && t == 0) estamp[2] = __builtin_readcyclecounter();
#endif
                if (!h1) {
#pragma unroll
                    for (int k = 0; k < OC; ++k) {
                        tot[k] = epi_affine(tot[k], eR[k], p.useBN);
                    }
                    if (actSimple) { // (tested once per row tile, not per value: a branch is a pipeline drain)
#pragma unroll
                        for (int k = 0; k < OC; ++k) tot[k] = __builtin_amdgcn_fmed3f(fmaxf(tot[k], tot[k] * ac.alpha), ac.lo, ac.hi);
                    } else {
#pragma unroll
                        for (int k = 0; k < OC; ++k) tot[k] = epi_act(ac.act, ac.leaky, tot[k], 0.0f);
                    }
#pragma unroll
                    for (int k = 0; k < OC; ++k) {
                        line[(t * 32 + l32) * OC + k] = static_cast<_Float16>(tot[k]);
                    }
                }
            }
#endif
#ifdef SNNHIP_RM_TRACE
            if (rtrace && j == 0) estamp[3] = __builtin_readcyclecounter();
#endif
            wave_lds_sync();
#ifdef SNNHIP_RM_TRACE
            if (rtrace && j == 0) estamp[4] = __builtin_readcyclecounter();
#endif
            // the row: validCols * OC halfs, contiguous in the output
            const size_t base = (static_cast<size_t>(n) * p.OH + oy) * rowHalfs + static_cast<size_t>(ox0) * OC;
            _Float16* const yr = y + base;
            const int nh = validCols * OC;
            if ((base & 1) == 0) { // (uniform) 4-byte aligned row start: consecutive lanes store consecutive dwords
                const unsigned* const lw = reinterpret_cast<const unsigned*>(line);
                unsigned* const yw = reinterpret_cast<unsigned*>(yr);
                constexpr int NQ = (TW * OC / 2 + 63) / 64;
                unsigned dw[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) dw[q] = lw[lane + 64 * q]; // (both reads first: one LDS round trip per row, not one per store)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int d = lane + 64 * q;
                    if (2 * d + 1 < nh) yw[d] = dw[q];
                }
                if ((nh & 1) && lane == 0) yr[nh - 1] = line[nh - 1];
            } else {
#pragma unroll
                for (int q = 0; q < (TW * OC + 63) / 64; ++q) {
                    const int e = lane + 64 * q;
                    if (e < nh) yr[e] = line[e];
                }
            }
            asm volatile("" ::: "memory"); // (the line of row j is not written again before the next iteration: LDS order does the rest)
#ifdef SNNHIP_RM_TRACE
            if (rtrace && j == 0) estamp[5] = __builtin_readcyclecounter();
#endif
        }

        RM_MARK(4);
        if (more) store_group(it + 2);
        RM_MARK(5);
        __syncthreads();
#ifdef SNNHIP_RM_TRACE
        if (rtrace && it >= 4 && it < 7)
            printf("rmepi tid %d it %d: j0t0 pulls+acc %llu shfl %llu | row0 to-line %llu sync %llu stores %llu\n", tid, it, estamp[1] - estamp[0], estamp[2] - estamp[1], estamp[3] - estamp[0],
                   estamp[4] - estamp[3], estamp[5] - estamp[4]);
        if (rtrace && it >= 4 && it < 7)
            printf("rmtrace tid %d it %d: loads %llu mfma %llu bar1 %llu epi %llu batch %llu bar2 %llu total %llu\n", tid, it, rstamp[1] - rstamp[0], rstamp[2] - rstamp[1], rstamp[3] - rstamp[2],
                   rstamp[4] - rstamp[3], rstamp[5] - rstamp[4], __builtin_readcyclecounter() - rstamp[5], __builtin_readcyclecounter() - rstamp[0]);
#endif
    }
}

struct RowmarchPlan : ConvPlanBase {
    RowmarchParams p;
    ActCfg ac;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    size_t ldsBytes = 0;
    dim3 grid;
    void (*kernel)(RowmarchParams, ActCfg, const _Float16*, const float4*, const float4*, _Float16*) = nullptr;

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "conv2d: expects 1 input, got %d", nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.srcH && x->w == p.srcW && x->c == p.IC, "conv2d: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h,
                       x->w, x->c, p.N, p.srcH, p.srcW, p.IC);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.OC, "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n,
                       out->h, out->w, out->c, p.N, p.OH, p.OW, p.OC);
        SNNHIP_LAUNCH(kernel, grid, dim3(256), ldsBytes, ctx->stream, p, ac, reinterpret_cast<const _Float16*>(x->data), reinterpret_cast<const float4*>(d_w),
                           reinterpret_cast<const float4*>(d_epi), reinterpret_cast<_Float16*>(out->data));
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

typedef void (*RowmarchFn)(RowmarchParams, ActCfg, const _Float16*, const float4*, const float4*, _Float16*);
template <int K, int ICS>
RowmarchFn pick_oc(int oc) {
    switch (oc) {
    case 1: return conv2d_rowfold_march_kernel<K, ICS, 1>;
    case 2: return conv2d_rowfold_march_kernel<K, ICS, 2>;
    case 3: return conv2d_rowfold_march_kernel<K, ICS, 3>;
    default: return (K * 4 <= 32) ? conv2d_rowfold_march_kernel<K, ICS, (K * 4 <= 32 ? 4 : 1)> : nullptr;
    }
}
template <int K>
RowmarchFn pick_rowmarch(int ics, int oc) {
    return ics == 1 ? pick_oc<K, 1>(oc) : pick_oc<K, 2>(oc);
}

} // namespace

// Tried first by make_conv2d_rowfold_plan (conv2d_rowfold.hip); SNNHIP_E_UNSUPPORTED hands the layer to the tile kernel there.
int make_conv2d_rowmarch_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    const char* form = snnhip::option("SNNHIP_ROWFOLD");
    if (form && strcmp(form, "tile") == 0) return SNNHIP_E_UNSUPPORTED;
    if (g.normShift && !act_is_simple(g.normAct)) return SNNHIP_E_UNSUPPORTED;
    if (g.dtype != SNNHIP_F16 || g.kh != g.kw || (g.kh != 5 && g.kh != 7 && g.kh != 9) || g.sh != 1 || g.sw != 1) return SNNHIP_E_UNSUPPORTED;
    if (g.OC > 4 || g.kh * g.OC > 32 || (g.IC != 16 && g.IC != 32) || g.act == SNNHIP_ACT_SILU_QUIRK || g.addAct >= 0) return SNNHIP_E_UNSUPPORTED;
    if (static_cast<double>(g.N) * g.H * g.W * g.IC >= 2147483647.0) return SNNHIP_E_UNSUPPORTED;
    const int K = g.kh, ICS = g.IC / 16, TW = kCols - K + 1;
    RowmarchFn fn = K == 9 ? pick_rowmarch<9>(ICS, g.OC) : K == 7 ? pick_rowmarch<7>(ICS, g.OC) : pick_rowmarch<5>(ICS, g.OC);
    if (!fn) return SNNHIP_E_UNSUPPORTED;
    RowmarchParams p = {};
    p.N = g.N; p.H = g.H; p.W = g.W; p.IC = g.IC; p.OC = g.OC; p.OH = g.OH; p.OW = g.OW; p.padx = g.padx; p.pady = g.pady; p.padMode = g.padMode; p.useBN = g.useBN;
    p.preMode = g.preMode; p.preX = g.preX; p.preY = g.preY; p.preShift = g.preShift;
    p.srcH = g.preMode ? g.srcH : g.H;
    p.srcW = g.preMode ? g.srcW : g.W;
    p.tilesX = up_div(g.OW, TW);
    // row segments: a strip is split so that the grid fills whole rounds of the chip's 2 x CUs block slots, the 8 halo rows a segment re-reads
    // weighed against the tail of a partly filled last round
    {
        const int slots = 2 * std::max(1, ctx->props.multiProcessorCount), strips = g.N * p.tilesX;
        int bestSegs = 1;
        double bestEff = -1.0;
        const char* forced = snnhip::option("SNNHIP_ROWFOLD_SEGS");
        for (int s = 1; s <= 64; ++s) {
            const int rows = round_up(up_div(g.OH, s), kTH);
            const int segs = up_div(g.OH, rows);
            if (segs != s) continue; // (this s rounds to a segment count already tried)
            if (s > 1 && rows < 3 * kTH && !forced) break;
            const double blocks = static_cast<double>(strips) * segs;
            const double eff = blocks / (std::ceil(blocks / slots) * slots) * rows / (rows + kTH);
            if ((forced && atoi(forced) == s) || (!forced && eff > bestEff + 1e-9)) {
                bestEff = eff;
                bestSegs = segs;
                if (forced) break;
            }
        }
        p.segs = bestSegs;
        p.segRows = round_up(up_div(g.OH, bestSegs), kTH);
        p.segs = up_div(g.OH, p.segRows);
    }
    p.normShift = g.normShift; p.normMul = g.normMul;
    p.normAc = make_act_cfg(g.normShift ? g.normAct : SNNHIP_ACT_NONE, g.normLeaky);
    const size_t lds = static_cast<size_t>(16) * kCols * g.IC * 2 + 4 * 2 * kCols * 4 * sizeof(_Float16);
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
        set_error("conv2d_rowmarch: hipFuncSetAttribute(%zu) failed", lds);
        return SNNHIP_E_HIP;
    }
    auto* plan = new RowmarchPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * K * K);
    plan->epi4 = epi4;
    plan->p = p;
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->kernel = fn;
    plan->ldsBytes = lds;
    plan->grid = dim3(static_cast<unsigned>(p.tilesX) * p.segs * g.N);
    plan->dtype = SNNHIP_F16;
    // weights: Wp[step = fy * ICS + c][lane = 32 h + n] x 8 halfs {W[oc][16 c + 8 h + j][fy][fx]}, n = fx * OC + oc; rows >= K * OC are zero
    // (conv2d_rowfold.hip's pack: the 32x32x16 MFMA's A and B operand layouts are the same)
    const int NK = K * ICS;
    std::vector<float> wpk(static_cast<size_t>(NK) * 64 * 4, 0.0f);
    _Float16* wph = reinterpret_cast<_Float16*>(wpk.data());
    for (int oc = 0; oc < g.OC; ++oc)
        for (int ic = 0; ic < g.IC; ++ic)
            for (int fy = 0; fy < K; ++fy)
                for (int fx = 0; fx < K; ++fx) {
                    const int c = ic / 16, hh = (ic % 16) / 8, j = ic % 8, nn = fx * g.OC + oc;
                    wph[((static_cast<size_t>(fy) * ICS + c) * 64 + hh * 32 + nn) * 8 + j] =
                        static_cast<_Float16>(w_oihw[((static_cast<size_t>(oc) * g.IC + ic) * K + fy) * K + fx]);
                }
    int rc = plan->upload(wpk.data(), wpk.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epi4.data(), epi4.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = g.N; plan->inDims[1] = p.srcH; plan->inDims[2] = p.srcW; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->flops = 2.0 * K * K * g.IC * g.OC * static_cast<double>(g.OH) * g.OW * g.N;
    plan->bytes = 2.0 * (static_cast<double>(g.N) * p.srcH * p.srcW * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC + static_cast<double>(g.OC) * g.IC * K * K);
    char buf[320];
    snprintf(buf, sizeof(buf), "conv2d_rowfold_mfma_f16_32x32x16 k=%dx%d s=1 ic=%d oc=%d (columns folded into N: %d of 32) row-marching strips=%dx%dpx segments=%d x %d rows lds=%zuB",
             K, K, g.IC, g.OC, K * g.OC, kTH, TW, p.segs, p.segRows, lds);
    plan->desc = buf;
    if (g.preMode) plan->desc += " +pad(" + std::string(g.preMode == SNNHIP_PAD_REFLECT ? "reflect" : g.preMode == SNNHIP_PAD_REPLICATE ? "replicate" : "constant") + ")";
    if (g.preMode && g.preShift) plan->desc += " +upsample(x2)";
    if (g.normShift) plan->desc = "instancenorm(act=" + std::to_string(g.normAct) + ", in the staging) -> " + plan->desc;
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
