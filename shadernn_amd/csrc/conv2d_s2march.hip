// conv2d_s2march.hip -- fp16 3x3 STRIDE-2 convolution for large feature maps with 32 or 64 input channels (the two down-sampling layers of the
// fast-neural-style networks, BASELINE configs[4]: 32 -> 64 @ 728 x 1288 and 64 -> 128 @ 364 x 644 per image), row-marching like conv2d_rowmarch.hip.
//
// Round 2 ran these layers on conv2d_mfma_kernel's 128-pixel blocks: 590 + 584 us per 16 images with the matrix pipe 10 % busy and 2x the input
// fetched (profiles/r02_c5_*) -- 36 MFMAs per wave behind a block's whole prologue / staging / epilogue, 29 000 blocks per launch.  A stride-2 layer
// has almost no halo to share (3 input rows per output row, 2 of them its own), so what a kernel can win here is not re-use but (a) weights that
// stay in registers for a whole strip instead of being streamed per 128 pixels, (b) the next rows in flight while the current ones are
// multiplied, (c) no per-tile prologue: one block = one 32-column strip of one image, marching down a segment of its rows.
//
//   * block = 512 threads = 8 waves = WR output rows x WN 32-channel column tiles (32 -> 64: 4 rows x 2 tiles, 64 -> 128: 2 rows x 4 tiles); per
//     iteration TH = WR output rows, i.e. 2 TH NEW input rows (+ 1 kept from the previous iteration) in an LDS ring of 2 TH + 1 rows;
//   * a wave = ONE output row x ONE 32-channel tile: all 9 x IC/16 weight operands of its tile in registers (the MFMA's A operand, 72 / 144
//     VGPRs), the 32 pixels of the row are the B operand (LDS: pixel pitch Q + 1 sixteen-byte slots -- odd, so the stride-2 operand reads are
//     conflict-free and every tap / channel-slot displacement is an immediate offset), 9 x IC/16 MFMAs per wave and iteration;
//   * the rows of iteration it + 1 are requested before the MFMAs of iteration it (registers), normalised (graph rule I) and written over the
//     rows that have just retired; the fused Pad in front (rule D) resolves in the row / column look-ups;
//   * the output tile (TH x 32 pixels x 32 WN channels) leaves through LDS as 16-byte channel-contiguous vectors.
// Operator contract as the other convolution kernels (shadertemplate_vk_conv2d.comp:148-347: padding modes, bias -> BN -> activation).
#include <cstring>
#include <vector>

#include "epilogue.h"
#include "norm_fold.h"
#include "snnhip_internal.h"

#include <type_traits>

namespace snnhip {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

struct S2Params {
    int N, H, W, IC, OC, OH, OW, padx, pady, padMode, useBN;
    int preMode, preX, preY, srcH, srcW, preShift; // fused Pad / nearest x2 upsampling in front (ConvGeom)
    int tilesX, segs, segRows;                     // column strips, row segments per strip, output rows per segment (multiple of TH)
    const float* normShift;                        // InstanceNorm in front (graph rule I); null = none
    const float* normMul;
    ActCfg normAc;
    // chain rule F (Conv2D -> InstanceNorm), as in conv2d_upconv.hip: one record {pixels, sum (v - bias), sum (v - bias)^2} per block and channel,
    // statRec[((n * gridDim.y + by) * blocksPerImage + block)][1 + 2 BN]; the last block of an image folds them into the norm's shift / mul.  null = off
    float* statRec;
    NormFoldArgs fold;
};

// (a 32-column output strip stages input columns 0 .. 64: 2 * 31 + 3 of them)
// The pixels of a ring row are DE-INTERLEAVED BY COLUMN PARITY: even columns 0, 2, .. 64 at pixel slots 0 .. 32, odd columns 1, 3, .. 63 at kOdd ..
// kOdd + 31.  A stride-2 operand read (lane l32 -> column 2 l32 + fx) then walks CONSECUTIVE pixel slots of one plane: with the odd pixel pitch of
// Q + 1 sixteen-byte slots the 16 lanes of a ds_read_b128 lane group hit 16 different bank slots.  (Interleaved, lane l32 sat at slot 2 (Q + 1) l32:
// even slots only -- PMC of the first version: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE.)  kOdd * (Q + 1) = 4 (mod 8) keeps the eight
// lanes of a staging ds_write_b128 group (two pixels x 4 slots for IC = 32) on distinct slots too.
#ifdef SNNHIP_S2_TRACE // experiment builds (tools/exp_one.sh): block 0 prints the s_memtime stamps of its phases for a few iterations
#define S2_MARK(i) do { if (trace) tstamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define S2_MARK(i) do { } while (0)
#endif
constexpr int kOdd = 36;
constexpr int kRowPix = kOdd + 32; // pixel slots per ring row
__host__ __device__ constexpr int pix_slot(int c) { return (c & 1) ? kOdd + (c >> 1) : (c >> 1); }

template <int ICS /* IC / 16: 2 | 4 */, int WN /* 32-channel tiles per block: 2 | 4 */, int PF /* batches of rows in flight: 1 | 2 */>
__global__ __launch_bounds__(512, 2) void conv2d_s2march_kernel(S2Params p, ActCfg ac, const _Float16* __restrict__ x, const float4* __restrict__ wp,
                                                                const float4* __restrict__ epi, _Float16* __restrict__ y) {
    constexpr int Q = 2 * ICS;          // 16-byte slots per pixel
    constexpr int QP = Q + 1;           // ... and its pitch in LDS
    constexpr int WR = 8 / WN;          // output rows per iteration (one per wave row)
    constexpr int TH = WR;
    constexpr int GR = 2 * TH;          // new input rows per iteration
    constexpr int RING = GR + 1;        // ring rows
    constexpr int ROWF = kRowPix * QP * 4;  // floats per ring row
    constexpr int EPR = 64 * Q;         // main staging elements per row (256 | 512)
    constexpr int RPR = 512 / EPR;      // rows per staging round (2 | 1)
    constexpr int NRND = GR / RPR;      // rounds per batch of GR rows (4)
    constexpr int BN = 32 * WN;         // output channels per block
    constexpr int EP = BN + 8;          // halfs per pixel of the output tile in LDS
    static_assert(NRND * RPR == GR && GR * Q == 32, "staging map");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* const otile = reinterpret_cast<_Float16*>(smem + RING * ROWF);                         // [TH][32][EP] halfs
    float* const normTab = smem + RING * ROWF + (TH * 32 * EP) / 2;                                  // [2][IC] shift, mul of this block's image
    float* const epiTab = normTab + 2 * 64;                                                          // [2][BN] scale, shift of this block's channels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int wr = wave / WN, wn = wave % WN;
    const int bx = blockIdx.x;
    const int seg = bx % p.segs, tx = (bx / p.segs) % p.tilesX, n = bx / (p.segs * p.tilesX);
    const int ox0 = tx * 32, oyS = seg * p.segRows, oyE = min(p.OH, oyS + p.segRows);
    const int ix0 = 2 * ox0 - p.padx, iyS = 2 * oyS - p.pady;
    const int nIter = (oyE - oyS + TH - 1) / TH;
    const int ocb = blockIdx.y * BN;

    // ---- this wave's weights (the MFMA's A operand): wq[tap][cc] = 8 halfs {W[ocb + 32 wn + l32][16 cc + 8 h + j][fy][fx]}
    float4 wq[9][ICS];
    {
        const float4* wt = wp + (static_cast<size_t>(blockIdx.y * WN + wn) * 9 * ICS) * 64 + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int cc = 0; cc < ICS; ++cc) wq[t][cc] = wt[(t * ICS + cc) * 64];
    }
    // epilogue folded to one fma per value, act(acc * scale + shift): scale = bnScale (1 without BN), shift = bnScale * (bias - bnMean) + bnBeta (bias).
    // The table sits in LDS: read per channel from global memory inside the loop (16 dependent loads per iteration, each behind an s_waitcnt vmcnt(0)
    // that also drained the row prefetch) the first version spent ~4 us per iteration on it -- its whole run time.
    if (tid < BN) {
        const float4 e4 = epi[ocb + tid];
        epiTab[tid] = p.useBN ? e4.y : 1.0f;
        epiTab[BN + tid] = p.useBN ? fmaf(e4.y, e4.x - e4.z, e4.w) : e4.x;
    }
    if (p.normShift)
        for (int i = tid; i < p.IC; i += 512) {
            normTab[i] = p.normShift[n * p.IC + i];
            normTab[p.IC + i] = p.normMul[n * p.IC + i];
        }

    // ---- staging map: thread -> column c (0..63) and slot sl of row r * RPR + rsub of a batch; threads 0..31 also carry column 64 (row tid / Q)
    const _Float16* xn = x + static_cast<size_t>(n) * p.srcH * p.srcW * p.IC;
    const int sl = tid % Q, c = (tid / Q) % 64;
    const int rsub = __builtin_amdgcn_readfirstlane(tid / EPR);
    auto resolve_col = [&](int cc) {
        int sx = resolve_nobranch(ix0 + cc, p.W, p.padMode);
        const int px = resolve_nobranch(sx - p.preX, p.srcW << p.preShift, p.preMode);
        const int pre = sx < 0 ? -1 : (px < 0 ? -1 : px >> p.preShift);
        return p.preMode ? pre : sx;
    };
    const int sxMain = resolve_col(c), sxLast = resolve_col(64);
    const int colOfs = max(sxMain, 0) * p.IC + 8 * sl, colOfsLast = max(sxLast, 0) * p.IC + 8 * sl;
    const int ldsMain = (pix_slot(c) * QP + sl) * 4, ldsLast = (pix_slot(64) * QP + sl) * 4; // + ring row * ROWF
    const int rowLast = tid / Q;                                          // (threads 0..31) row of the batch their column-64 element sits in
    const bool nRelu = p.normAc.act == SNNHIP_ACT_RELU;

    // a batch of GR rows in flight: the thread's NRND elements (+ its column-64 element), which of their rows exist, the ring rows they go to
    struct Batch {
        float4 v[NRND], vLast;
        unsigned rowOkMask;
        bool lastOk;
        int ringOf[NRND], ringLast;
    };
    // batch b = relative input rows GR b + 1 .. GR b + GR of the segment (b = -1: the rows up to row 0)
    auto load_batch = [&](int b, Batch& B) {
        const int r0 = GR * b + 1;
        int syv = resolve_nobranch(iyS + r0 + (lane & 7), p.H, p.padMode); // lane l resolves row l of the batch once, on the vector unit
        {
            const int py = resolve_nobranch(syv - p.preY, p.srcH << p.preShift, p.preMode);
            const int pre = syv < 0 ? -1 : (py < 0 ? -1 : py >> p.preShift);
            syv = p.preMode ? pre : syv;
        }
        B.rowOkMask = 0;
#pragma unroll
        for (int r = 0; r < NRND; ++r) {
            const int rr = r * RPR + rsub; // (wave-uniform)
            const int sy = __builtin_amdgcn_readlane(syv, rr);
            B.rowOkMask |= static_cast<unsigned>(sy >= 0) << r;
            B.ringOf[r] = ((r0 + rr) % RING + RING) % RING;
            B.v[r] = *reinterpret_cast<const float4*>(xn + static_cast<size_t>(max(sy, 0)) * p.srcW * p.IC + colOfs);
        }
        if (tid < 32) {
            const int sy = __shfl(syv, rowLast); // (lanes 0..31 of wave 0)
            B.lastOk = sy >= 0;
            B.ringLast = ((r0 + rowLast) % RING + RING) % RING;
            B.vLast = *reinterpret_cast<const float4*>(xn + static_cast<size_t>(max(sy, 0)) * p.srcW * p.IC + colOfsLast);
        }
    };
    // graph rule I on 8 staged channels 8 sl ..: half(act(x * mul + shift)) in fp32, the norm sweep's own arithmetic.  The thread's channel slot never
    // changes: with 32 input channels its 8 shifts and 8 multipliers stay in registers (the 64-channel instantiation has none to spare and reads them
    // from the LDS table per element, four ds_read_b128); and the activation kind is tested once per batch, not selected per value (the select
    // computed both forms: five instructions a value where ReLU needs one).  Phase trace, round 4: this pass was 3 200-3 900 of an iteration's 7 000 cycles.
    constexpr bool kNormRegs = ICS == 2;
    float nSh[8], nMu[8];
    if (kNormRegs && p.normShift) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            nSh[k] = p.normShift[n * p.IC + 8 * sl + k];
            nMu[k] = p.normMul[n * p.IC + 8 * sl + k];
        }
    }
    auto normalise = [&](float4& q, auto reluTag) {
        h8 hv = *reinterpret_cast<const h8*>(&q);
        const float* tb = normTab + 8 * sl;
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
            float shv[4], muv[4];
            if (kNormRegs) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    shv[k] = nSh[4 * q4 + k];
                    muv[k] = nMu[4 * q4 + k];
                }
            } else {
                const float4 sh = *reinterpret_cast<const float4*>(tb + 4 * q4), mu = *reinterpret_cast<const float4*>(tb + p.IC + 4 * q4);
                shv[0] = sh.x; shv[1] = sh.y; shv[2] = sh.z; shv[3] = sh.w;
                muv[0] = mu.x; muv[1] = mu.y; muv[2] = mu.z; muv[3] = mu.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float f = fmaf(static_cast<float>(hv[4 * q4 + k]), muv[k], shv[k]);
                hv[4 * q4 + k] = static_cast<_Float16>(decltype(reluTag)::value ? fmaxf(f, 0.0f) : __builtin_amdgcn_fmed3f(fmaxf(f, f * p.normAc.alpha), p.normAc.lo, p.normAc.hi));
            }
        }
        q = *reinterpret_cast<const float4*>(&hv);
    };
    auto store_batch = [&](Batch& B) { // normalise and write a landed batch into its ring rows; padding stays zero
        if (p.normShift) { // (uniform)
            if (nRelu) {
#pragma unroll
                for (int r = 0; r < NRND; ++r) normalise(B.v[r], std::true_type{});
                if (tid < 32) normalise(B.vLast, std::true_type{});
            } else {
#pragma unroll
                for (int r = 0; r < NRND; ++r) normalise(B.v[r], std::false_type{});
                if (tid < 32) normalise(B.vLast, std::false_type{});
            }
        }
#pragma unroll
        for (int r = 0; r < NRND; ++r) {
            const bool live = ((B.rowOkMask >> r) & 1u) && sxMain >= 0;
            const float4 o = make_float4(live ? B.v[r].x : 0.f, live ? B.v[r].y : 0.f, live ? B.v[r].z : 0.f, live ? B.v[r].w : 0.f);
            *reinterpret_cast<float4*>(smem + B.ringOf[r] * ROWF + ldsMain) = o;
        }
        if (tid < 32) {
            const bool live = B.lastOk && sxLast >= 0;
            const float4 o = make_float4(live ? B.vLast.x : 0.f, live ? B.vLast.y : 0.f, live ? B.vLast.z : 0.f, live ? B.vLast.w : 0.f);
            *reinterpret_cast<float4*>(smem + B.ringLast * ROWF + ldsLast) = o;
        }
    };

    // ---- MFMA B operand (pixels): lane (l32, h) reads column 2 l32 + fx = pixel slot l32 (fx = 0), kOdd + l32 (fx = 1), l32 + 1 (fx = 2), channel
    // slot 2 cc + h of a ring row: the lane part of the float offset (the tap part is an immediate)
    const int bofs = (l32 * QP + h) * 4;

    Batch b0, b1;
    __syncthreads(); // the norm table
    load_batch(-1, b0);
    store_batch(b0);
    __syncthreads(); // (batch 0 overwrites the ring rows batch -1 used for the rows in front of the segment, from other threads)
    load_batch(0, b0);
    store_batch(b0);
    // PF = 2: TWO batches stay in flight (the first version, one batch = 33 KB per CU, ran at the latency of its loads: ~5 us per iteration whatever
    // the iteration computed); batch it + 1 was requested an iteration ago and is written to the ring now, batch it + 2 is requested first thing
    if (PF == 2 && nIter > 1) load_batch(1, b1);
    __syncthreads();

    const float* const et = epiTab + 32 * wn + 4 * h; // this lane's channel runs: 8 g + 4 h + k of the wave's 32-channel tile
    const bool actSimple = act_is_simple_dev(ac.act);
    // rule F: a thread's vectors of the store loop are pixels of ONE 8-channel column: sums around the channel's bias of the stored (rounded) values
    float st1[8], st2[8], stN = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) st1[k] = st2[k] = 0.0f;
    // X: the batch written to the ring at the end of this iteration (rows of iteration it + 1); Y (PF = 2): the buffer the request for it + 2 goes to
#ifdef SNNHIP_S2_TRACE
    const bool trace = blockIdx.x == 3 && blockIdx.y == 0 && (tid == 0 || tid == 448);
    unsigned long long tstamp[8] = {};
#endif
    auto iteration = [&](int it, Batch& X, Batch& Y) {
        const bool more = it + 1 < nIter;
        S2_MARK(0);
        if (PF == 2) {
            if (it + 2 < nIter) load_batch(it + 2, Y);
        } else if (more) {
            load_batch(it + 1, X);
        }

        S2_MARK(1);
        // ---- wave = output row TH it + wr: input rows GR it + 2 wr + fy
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        const int r0 = __builtin_amdgcn_readfirstlane(GR * it + 2 * wr);
        // operand registers: a[cc] is dead once its MFMA has issued and is refilled for the next tap at once (the other MFMAs of the tap cover the
        // LDS round trip) -- one set of ICS operands, not two
        float4 a[ICS];
        auto tap_ptr = [&](int t) { return smem + ((r0 + t / 3) % RING) * ROWF + bofs + pix_slot(t % 3) * QP * 4; }; // tap t = 3 fy + fx
        {
            const float* rowp = tap_ptr(0);
#pragma unroll
            for (int cc = 0; cc < ICS; ++cc) a[cc] = *reinterpret_cast<const float4*>(rowp + 2 * cc * 4);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* nextp = tap_ptr(t + 1 < 9 ? t + 1 : t);
#pragma unroll
            for (int cc = 0; cc < ICS; ++cc) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&wq[t][cc]), *reinterpret_cast<const h8*>(&a[cc]), acc, 0, 0, 0);
                if (t + 1 < 9) a[cc] = *reinterpret_cast<const float4*>(nextp + 2 * cc * 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        S2_MARK(2);
        // ---- epilogue into the LDS tile: acc[4 g + k] = channel ocb + 32 wn + 8 g + 4 h + k of pixel l32 of row wr
        // (the phase trace of the first version: 2950 of an iteration's 6400 cycles sat HERE -- a table read, a wait and a branch on the activation
        // kind per value.  All table reads are issued first, and the activation kind is tested once per iteration, outside the value loops.)
        float4 sh4[4], sc4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) sh4[g] = *reinterpret_cast<const float4*>(et + BN + 8 * g);
        if (p.useBN) {
#pragma unroll
            for (int g = 0; g < 4; ++g) sc4[g] = *reinterpret_cast<const float4*>(et + 8 * g);
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) sc4[g] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        }
        float rv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float scv[4] = {sc4[g].x, sc4[g].y, sc4[g].z, sc4[g].w}, shv[4] = {sh4[g].x, sh4[g].y, sh4[g].z, sh4[g].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) rv[4 * g + k] = fmaf(acc[4 * g + k], scv[k], shv[k]);
        }
        if (actSimple) {
#pragma unroll
            for (int i = 0; i < 16; ++i) rv[i] = __builtin_amdgcn_fmed3f(fmaxf(rv[i], rv[i] * ac.alpha), ac.lo, ac.hi);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) rv[i] = epi_act(ac.act, ac.leaky, rv[i], 0.0f);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            h4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = static_cast<_Float16>(rv[4 * g + k]);
            *reinterpret_cast<h4*>(otile + (wr * 32 + l32) * EP + 32 * wn + 8 * g + 4 * h) = o;
        }
        S2_MARK(3);
        __syncthreads(); // the tile is complete, and every wave is done with the ring rows that retire
        S2_MARK(4);

        // ---- the tile leaves as 16-byte vectors, a pixel's BN channels contiguous
#pragma unroll
        for (int q = 0; q < (TH * 32 * (BN / 8)) / 512; ++q) {
            const int vi = tid + 512 * q;
            const int pix = vi / (BN / 8), c8 = vi % (BN / 8);
            const int oy = oyS + it * TH + (pix >> 5), ox = ox0 + (pix & 31);
            if (oy < oyE && ox < p.OW) {
                const float4 ov = *reinterpret_cast<const float4*>(otile + pix * EP + 8 * c8);
                *reinterpret_cast<float4*>(y + ((static_cast<size_t>(n) * p.OH + oy) * p.OW + ox) * p.OC + ocb + 8 * c8) = ov;
                if (ICS == 2 && p.statRec) { // (uniform; the 64-channel form has no registers left for the accumulators and is never asked)
                    const h8 hv = *reinterpret_cast<const h8*>(&ov);
                    const float4 pa = *reinterpret_cast<const float4*>(epiTab + BN + 8 * c8), pb = *reinterpret_cast<const float4*>(epiTab + BN + 8 * c8 + 4);
                    const float pv[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
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
        S2_MARK(5);
        if (more) store_batch(X);
        S2_MARK(6);
        __syncthreads();
        S2_MARK(7);
#ifdef SNNHIP_S2_TRACE
        if (trace && it >= 4 && it < 8)
            printf("s2trace tid %d it %d: loads %llu mfma %llu epi %llu bar1 %llu stores %llu batch %llu bar2 %llu total %llu\n", tid, it, tstamp[1] - tstamp[0],
                   tstamp[2] - tstamp[1], tstamp[3] - tstamp[2], tstamp[4] - tstamp[3], tstamp[5] - tstamp[4], tstamp[6] - tstamp[5], tstamp[7] - tstamp[6], tstamp[7] - tstamp[0]);
#endif
    };
    if (PF == 2) {
        for (int it = 0; it < nIter; it += 2) { // (two iterations per trip: the batch buffers alternate without run-time register indexing)
            iteration(it, b1, b0);
            if (it + 1 < nIter) iteration(it + 1, b0, b1);
        }
    } else {
        for (int it = 0; it < nIter; ++it) iteration(it, b0, b0);
    }

    if (ICS != 2 || !p.statRec) return; // (uniform)
    // ---- the block's record, the count of finished blocks, and in the image's last block the fold (conv2d_upconv.hip)
    {
        constexpr int T = 512, CPT = BN / 8, TPC = T / CPT, PARTS = T / BN;
        float* const red = smem; // [17][T]: the ring and the tile are dead
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[k * T + tid] = st1[k];
            red[(8 + k) * T + tid] = st2[k];
        }
        red[16 * T + tid] = stN;
        __syncthreads();
        const int BPI = p.segs * p.tilesX;
        float* const rec = p.statRec + (static_cast<size_t>(n * gridDim.y + blockIdx.y) * BPI + (bx - n * BPI)) * (1 + 2 * BN);
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
            if (tid == 0) st_agent(rec, an);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned* cnt = p.fold.counter + n * gridDim.y + blockIdx.y;
            const unsigned prev = atomicAdd(cnt, 1u);
            const bool last = prev + 1u == static_cast<unsigned>(BPI);
            if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            red[0] = last ? 1.0f : 0.0f;
        }
        __syncthreads();
        if (red[0] == 0.0f) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int ch = tid % BN, part = tid / BN;
        const float* r0 = p.statRec + static_cast<size_t>(n * gridDim.y + blockIdx.y) * BPI * (1 + 2 * BN);
        float a1 = 0.0f, a2 = 0.0f, an = 0.0f;
        for (int b0i = part; b0i < BPI; b0i += 4 * PARTS) {
            float t1[4], t2[4], tn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = b0i + j * PARTS;
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
        __syncthreads();
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

struct S2marchPlan : ConvPlanBase {
    S2Params p;
    ActCfg ac;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    size_t ldsBytes = 0;
    dim3 grid;
    void (*kernel)(S2Params, ActCfg, const _Float16*, const float4*, const float4*, _Float16*) = nullptr;

    // chain rule F: per-block records, offered only together with the in-kernel fold (conv2d_upconv.hip)
    bool enableTileStats() override {
        if (statPart) return true;
        if (snnhip::option("SNNHIP_NO_KERNEL_FOLD") || p.IC != 32) return false;
        const int BPI = p.segs * p.tilesX, BN = static_cast<int>(p.OC / grid.y);
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
        SNNHIP_REQUIRE(!p.statRec || p.fold.counter, "conv2d_s2march: block statistics were switched on without the in-kernel fold (no fold launch reads its records)");
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.srcH && x->w == p.srcW && x->c == p.IC, "conv2d: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h,
                       x->w, x->c, p.N, p.srcH, p.srcW, p.IC);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.OC, "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n,
                       out->h, out->w, out->c, p.N, p.OH, p.OW, p.OC);
        SNNHIP_LAUNCH(kernel, grid, dim3(512), ldsBytes, ctx->stream, p, ac, reinterpret_cast<const _Float16*>(x->data), reinterpret_cast<const float4*>(d_w),
                           reinterpret_cast<const float4*>(d_epi), reinterpret_cast<_Float16*>(out->data));
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

} // namespace

// Tried by make_conv2d_mfma_plan (conv2d_mfma.hip) in front of the 128-pixel kernel; SNNHIP_E_UNSUPPORTED hands the layer on.
int make_conv2d_s2march_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    const char* force = snnhip::option("SNNHIP_CONV");
    const bool forced = force && strcmp(force, "s2march") == 0;
    if (g.normShift && !act_is_simple(g.normAct)) return SNNHIP_E_UNSUPPORTED;
    if (g.dtype != SNNHIP_F16 || g.kh != 3 || g.kw != 3 || g.sh != 2 || g.sw != 2 || (g.IC != 32 && g.IC != 64)) return SNNHIP_E_UNSUPPORTED;
    const int ICS = g.IC / 16, WN = g.IC == 32 ? 2 : 4, BN = 32 * WN, TH = 8 / WN;
    if (g.OC % BN != 0 || g.act == SNNHIP_ACT_SILU_QUIRK || g.addAct >= 0) return SNNHIP_E_UNSUPPORTED;
    if (static_cast<double>(g.N) * std::max(g.H, g.srcH) * std::max(g.W, g.srcW) * g.IC >= 2147483647.0) return SNNHIP_E_UNSUPPORTED;
    const int tilesX = up_div(g.OW, 32);
    // a strip is worth its prologue (2 batches of rows, the weights) from a few dozen output rows on, and the chip wants every CU busy
    if (!forced && (g.OH < 48 || static_cast<long>(g.N) * tilesX * up_div(g.OH, 48) < ctx->props.multiProcessorCount)) return SNNHIP_E_UNSUPPORTED;
    S2Params p = {};
    p.N = g.N; p.H = g.H; p.W = g.W; p.IC = g.IC; p.OC = g.OC; p.OH = g.OH; p.OW = g.OW; p.padx = g.padx; p.pady = g.pady; p.padMode = g.padMode; p.useBN = g.useBN;
    p.preMode = g.preMode; p.preX = g.preX; p.preY = g.preY; p.preShift = g.preShift;
    p.srcH = g.preMode ? g.srcH : g.H;
    p.srcW = g.preMode ? g.srcW : g.W;
    p.tilesX = tilesX;
    {   // row segments: whole rounds of one block per CU, the segment's prologue (GR rows re-read + the weights) weighed against the tail round
        const int slots = std::max(1, ctx->props.multiProcessorCount), strips = g.N * tilesX * (g.OC / BN);
        const char* fs = snnhip::option("SNNHIP_S2MARCH_SEGS");
        int bestSegs = 1;
        double bestEff = -1.0;
        for (int s = 1; s <= 64; ++s) {
            const int rows = round_up(up_div(g.OH, s), TH);
            const int segs = up_div(g.OH, rows);
            if (segs != s) continue;
            if (s > 1 && rows < 24 && !fs) break;
            const double blocks = static_cast<double>(strips) * segs;
            const double eff = blocks / (std::ceil(blocks / slots) * slots) * rows / (rows + 2 * TH + 2);
            if ((fs && atoi(fs) == s) || (!fs && eff > bestEff + 1e-9)) {
                bestEff = eff;
                bestSegs = segs;
                if (fs) break;
            }
        }
        p.segRows = round_up(up_div(g.OH, bestSegs), TH);
        p.segs = up_div(g.OH, p.segRows);
    }
    p.normShift = g.normShift; p.normMul = g.normMul;
    p.normAc = make_act_cfg(g.normShift ? g.normAct : SNNHIP_ACT_NONE, g.normLeaky);
    const int QP = 2 * ICS + 1, RING = 2 * TH + 1;
    const size_t lds = static_cast<size_t>(RING) * kRowPix * QP * 16 + static_cast<size_t>(TH) * 32 * (BN + 8) * 2 + 2 * 64 * 4 + 2 * static_cast<size_t>(BN) * 4;
    auto fn = g.IC == 32 ? conv2d_s2march_kernel<2, 2, 2> : conv2d_s2march_kernel<4, 4, 1>; // (64 input channels: 144 weight registers leave no room for a second batch)
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
        set_error("conv2d_s2march: hipFuncSetAttribute(%zu) failed", lds);
        return SNNHIP_E_HIP;
    }
    auto* plan = new S2marchPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * 9);
    plan->epi4 = epi4;
    plan->p = p;
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->kernel = fn;
    plan->ldsBytes = lds;
    plan->grid = dim3(static_cast<unsigned>(p.tilesX) * p.segs * g.N, static_cast<unsigned>(g.OC / BN));
    plan->dtype = SNNHIP_F16;
    // weights: Wp[32-channel tile nt][tap][cc][lane = 32 hh + m] x 8 halfs {W[32 nt + m][16 cc + 8 hh + j][fy][fx]}
    const int NT = g.OC / 32;
    std::vector<float> wpk(static_cast<size_t>(NT) * 9 * ICS * 64 * 4, 0.0f);
    _Float16* wph = reinterpret_cast<_Float16*>(wpk.data());
    for (int oc = 0; oc < g.OC; ++oc)
        for (int ic = 0; ic < g.IC; ++ic)
            for (int t = 0; t < 9; ++t) {
                const int nt = oc / 32, m = oc % 32, cc = ic / 16, hh = (ic % 16) / 8, j = ic % 8;
                wph[(((static_cast<size_t>(nt) * 9 + t) * ICS + cc) * 64 + hh * 32 + m) * 8 + j] = static_cast<_Float16>(w_oihw[(static_cast<size_t>(oc) * g.IC + ic) * 9 + t]);
            }
    int rc = plan->upload(wpk.data(), wpk.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epi4.data(), epi4.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = g.N; plan->inDims[1] = p.srcH; plan->inDims[2] = p.srcW; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->flops = 2.0 * 9 * g.IC * g.OC * static_cast<double>(g.OH) * g.OW * g.N;
    plan->bytes = 2.0 * (static_cast<double>(g.N) * p.srcH * p.srcW * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC + static_cast<double>(g.OC) * g.IC * 9);
    char buf[320];
    snprintf(buf, sizeof(buf), "conv2d_mfma_f16_32x32x16 k=3x3 s=2 ic=%d oc=%d row-marching strips=%dx32px x %doc (wave = 1 row x 32 oc, weights in registers) segments=%d x %d rows lds=%zuB",
             g.IC, g.OC, TH, BN, p.segs, p.segRows, lds);
    plan->desc = buf;
    if (g.preMode) plan->desc += " +pad(" + std::string(g.preMode == SNNHIP_PAD_REFLECT ? "reflect" : g.preMode == SNNHIP_PAD_REPLICATE ? "replicate" : "constant") + ")";
    if (g.preMode && g.preShift) plan->desc += " +upsample(x2)";
    if (g.normShift) plan->desc = "instancenorm(act=" + std::to_string(g.normAct) + ", in the staging) -> " + plan->desc;
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
