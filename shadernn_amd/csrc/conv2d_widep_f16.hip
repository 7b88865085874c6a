// conv2d_widep_f16.hip -- conv2d_wide_f16's 256-pixel x 128-channel block as a PERSISTENT kernel (round 5): fp16 Conv2D 3x3 stride 1, IC = 128, OC = 128 on
// large maps -- the ten body layers of the style-transfer graphs (Candy: 47 % of a step at 0.36 of the fp16 MFMA peak with one block per tile; 43 % at
// 0.415 with this kernel, 333 -> 292 us per launch inside the graph).
//
// What the one-block-per-tile kernel paid per tile besides its 576 MFMAs per wave (phase trace + ablations): a prologue of
// tables and staging maps (4 500 cycles), the first chunk's DMA in the open (4 300), an epilogue through a block-wide 68 KB LDS tile with three
// barriers (8 800), a statistics merge with three more, the image counter's atomic round trip in every block, and -- found while this kernel was
// written -- a drained prefetch at the head of every chunk: the compiler counts only ITS loads in s_waitcnt vmcnt(N), so with inline-asm LDS-DMA
// copies in the queue its counted wait for a weight operand also waited for the copies issued just before.
//
//   * grid = two blocks per CU; block b walks a CONTIGUOUS run of tiles (image-major, tilesBase / tilesRem), so it leaves an image once and its
//     statistics need one record per (block, image); 4 waves = 2 x 2 (pixel rows x 64-channel halves), a wave = 4 x 2 v_mfma_f32_32x32x16_f16 tiles;
//   * EVERY vector-memory instruction is inline assembly and every s_waitcnt vmcnt is counted by hand (the table at KSTEP below): weight operands
//     (a ring of D K-steps per lane, scalar base + 32-bit lane offset), LDS-DMA copies of the next chunk's halo tile and output stores through raw
//     buffer descriptors (a lane offset beyond the tensor = the copy writes zeros / the store is dropped: ragged tiles and zero padding cost no
//     branch and keep the instruction count the waits rely on).  The four chunks of a tile are unrolled (no loop back-edge carries a register a
//     load is still writing), the ring wraps from a tile's last steps into the next tile's first (the weights do not depend on the tile) and is
//     waited for once, under the epilogue's stores, before the tile loop's back-edge.  Two hazards the compiler's recogniser does not see in
//     inline assembly are padded by hand: a 16-byte store's data registers rewritten by the next VALU instruction (s_nop 1), and an SGPR written
//     by v_readfirstlane used as a buffer instruction's soffset (s_nop 4);
//   * everything that is not an MFMA is spread over the K-steps of the wave's OWN MFMA stream -- one copy per step in steps 0..6 of a chunk, the
//     next tile's tables at chunk 1 step 0, one element of its staging map per step of chunk 2, rule I's fix-up of the rows just copied in the
//     slots behind step 8.  Measured (phase traces, profiles/r05_widep_phase_trace_*.txt): the same VALU / LDS work as a phase of its own, beside
//     the partner wave's MFMAs, costs ~17 cycles per instruction; inside the wave's own K-steps it is almost free;
//   * the epilogue is WAVE-PRIVATE: a wave converts one 32-pixel row of its accumulators (sum + bias [-> BN] -> activation -> half), passes it
//     through its own 4.5 KB of LDS as 8-byte runs and reads it back as 16-byte vectors -- 8 lanes = the 128 contiguous bytes of a pixel's channel
//     half -- no block barrier, no 68 KB tile (so the block still fits twice on a CU beside its two 28 KB staging buffers);
//   * chain rule F: sums and squares around the channel's bias of the values a lane carries to memory, kept in registers across the tiles of
//     the run that lie in one image; when the run leaves the image the block writes ONE record (through to the coherence point), bumps the
//     image's counter, and the block that writes an image's last record folds it right there (norm_fold.h's hand-off).  Per-tile records with
//     a per-tile atomic were the first form: their acknowledgements sat in front of the weight loads in the in-order vmcnt queue;
//   * graph rule I (the InstanceNorm in front, applied in LDS behind the DMA) for none / ReLU in a two-instruction form; the other activations
//     of the norm stay on conv2d_wide_kernel.
// Semantics: shadertemplate_vk_conv2d.comp:148-347 (padding redirects :180-185,213-218), bit-identical to conv2d_wide_kernel's convolution (same
// K order); the statistics differ from its per-thread mean / M2 records only in how the same sums are grouped.
#include "conv2d_mfma_kernel.h"
#include "norm_fold.h"

#include <cstring>
#include <type_traits>

#ifdef SNNHIP_WIDEP_TRACE // experiment builds (tools/exp_one.sh): one block sums the s_memtime spans of its phases over its tiles and prints them
#define WP_T0() unsigned long long wpT = __builtin_readcyclecounter()
#define WP_ADD(i) do { const unsigned long long wpN = __builtin_readcyclecounter(); wpAcc[i] += wpN - wpT; wpT = wpN; } while (0)
#else
#define WP_T0() do { } while (0)
#define WP_ADD(i) do { } while (0)
#endif

namespace snnhip {

namespace {

using mfma_detail::f32x16;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int kTileW = 34;                        // staged tile: 32 + 2 columns
constexpr int kTileH = 10;                        // 8 + 2 rows
constexpr int kQP = 5;                            // 16-byte slots per staged pixel: 4 of data (32 channels) + 1 of padding (odd pitch: conflict-free ds_read_b128)
constexpr int kTotal = kTileH * kTileW * kQP;     // 1700 slots
constexpr int kR = 7;                             // DMA instructions per thread and chunk (7 x 256 >= 1700)
constexpr int kBufBytes = kR * 256 * 16;          // 28 672
constexpr int kScrPitch = 144;                    // bytes per pixel of a wave's epilogue scratch (64 halfs + 16: rows rotate by four banks)
constexpr int kScrBytes = 32 * kScrPitch;         // 4 608 per wave
constexpr int kLdsBuf0 = 0, kLdsBuf1 = kBufBytes, kLdsScr = 2 * kBufBytes, kLdsEpi = kLdsScr + 4 * kScrBytes /* float4[128] */, kLdsBias = kLdsEpi + 2048 /* float[128] */,
              kLdsSy = kLdsBias + 512 /* int[32] */, kLdsSx = kLdsSy + 128 /* int[40] */, kLdsNorm = kLdsSx + 160 /* [2 slots][2][IC] floats */;
constexpr int kStepBytes = 2 * 128 * 16;          // packed weights per K step: [h][oc] x 16 bytes
constexpr int kEpiStores = 16;                    // output stores per thread and tile (4 rows x 4 vectors)
constexpr int kRecFloats = 4 + 2 * 256;           // a statistics record: {pixels of wave row 0, of wave row 1, -, -}, then per wave row [S1[128] | S2[128]]

struct WidePParams {
    int N, H, W, OH, OW, padx, pady, padMode, useBN;
    unsigned tilesX, tilesY, tilesPerImage, numTiles;
    int preMode, preX, preY, srcH, srcW, preShift;
    unsigned xBytes, yBytes; // sizes of the input / output tensors (the raw buffer descriptors' bounds)
    // chain rule F: one record per (image, block that worked on it): [n][recsMax][kRecFloats] sums around the channels' biases; null = off.  A block's
    // tiles are a contiguous run, so it leaves one record per image it touched (rarely more than one) when its run leaves the image
    float* statRec;
    unsigned* counter;        // [N] records of the image written so far; the block that writes an image's last record folds it (norm_fold.h's hand-off)
    const NormFoldArgs* fold; // (device copy: read by the folding block only -- eleven scalar registers the tile loop does not have to carry)
    unsigned tilesBase, tilesRem, recsMax; // block b works on tiles [b base + min(b, rem), + base + (b < rem))
    const float* normShift; // graph rule I
    const float* normMul;
    ActCfg normAc;
};

template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// the wait for a weight operand: the registers go through the statement, so nothing that reads them can be scheduled in front of it
template <int N>
__device__ __forceinline__ void vm_wait_tie(f4& a, f4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
__device__ __forceinline__ void gload16x2(const char* sbase, unsigned voff, f4& r0, f4& r1) { // [oc tile 0, oc tile 1] of one K step (512 bytes apart)
    asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:512" : "=&v"(r0), "=&v"(r1) : "v"(voff), "s"(sbase));
}
// Tensors are addressed through raw buffer descriptors (base, 0 stride, size in bytes): a lane offset at or beyond the size is out of range -- a load
// returns zeros (the padding pixels of the staging map: no select against a block of zeros, no 64-bit address arithmetic per copy) and a store is
// dropped (pixels of a ragged tile outside the map: every thread still issues the same NUMBER of stores, which the counted waits rely on).
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    i4 r;
    r[0] = __builtin_amdgcn_readfirstlane(static_cast<int>(a));
    r[1] = __builtin_amdgcn_readfirstlane(static_cast<int>((a >> 32) & 0xffffu)); // stride 0, no swizzle
    r[2] = __builtin_amdgcn_readfirstlane(static_cast<int>(bytes));
    r[3] = 0x00020000; // gfx9 raw buffer: DATA_FORMAT 32
    return r;
}
constexpr unsigned kOutOfRange = 0xfffffff0u;
__device__ __forceinline__ unsigned in_sgpr(unsigned v) { // a scalar-register operand for inline assembly, also when the value is a compile-time constant
    v = __builtin_amdgcn_readfirstlane(v);                // (wave-uniform values the compiler sees as per-lane: a wave's row of the tile)
    asm volatile("s_nop 4" : "+s"(v));                    // (the MUBUF soffset field takes a register or an inline constant, not a literal; the wait states: a
                                                          // vector instruction's scalar result may not feed a vector-memory instruction for five cycles, and
                                                          // the hazard recognizer does not look inside inline assembly)
    return v;
}
__device__ __forceinline__ void lds_dma16_buf(const i4& rsrc, unsigned laneByteOffset, unsigned uniformByteOffset, unsigned ldsWaveByteAddr) {
    asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(laneByteOffset), "s"(rsrc), "s"(in_sgpr(ldsWaveByteAddr)), "s"(in_sgpr(uniformByteOffset)) : "memory");
}
// (s_nop 1: a store of more than 8 bytes reads its data registers over the following cycles; the compiler's hazard recognizer puts the wait states
// behind ITS stores, not behind an inline-asm one -- without them the address arithmetic of the next vector, allocated into the data's first register,
// reached memory in lanes 12-15 of every row of 16: found as wrong values at odd pixels, channels 32 u + 8 g + {0, 1})
__device__ __forceinline__ void store16_buf(const i4& rsrc, unsigned laneByteOffset, unsigned uniformByteOffset, const f4& v) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(laneByteOffset), "s"(rsrc), "s"(in_sgpr(uniformByteOffset)) : "memory");
}
__device__ __forceinline__ void lds_barrier() { // LDS traffic of this wave done, then the block barrier; vector memory stays in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The fold of an image's records (all 256 threads of the block that wrote the last one; scratch: 3 x 256 floats of LDS nobody uses).  Every record holds
// sums around the SAME pivots (the channels' biases), so adding them up in record order is the whole merge: deterministic whichever block folds.
// thread = (channel, 1 of 2 parts), four records (24 loads) in flight per thread; ~33 records per 720p image.
__device__ __forceinline__ void widep_stats_fold(const NormFoldArgs& f, const float* recs, int nRecs, const float* biasTab, float* scratch, int n) {
    const int tid = threadIdx.x, ch = tid & 127, part = tid >> 7;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // one acquire per image, in the folding block only (norm_fold.h)
    float an = 0.0f, a1 = 0.0f, a2 = 0.0f;
    for (int b0 = part; b0 < nRecs; b0 += 8) {
        float tn[4][2], t1[4][2], t2[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = b0 + 2 * j;
#pragma unroll
            for (int w = 0; w < 2; ++w) tn[j][w] = t1[j][w] = t2[j][w] = 0.0f;
            if (b < nRecs) {
                const float* rb = recs + static_cast<size_t>(b) * kRecFloats;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    tn[j][w] = ld_agent(rb + w);
                    t1[j][w] = ld_agent(rb + 4 + w * 256 + ch);
                    t2[j][w] = ld_agent(rb + 4 + w * 256 + 128 + ch);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                an += tn[j][w];
                a1 += t1[j][w];
                a2 += t2[j][w];
            }
    }
    scratch[tid] = an;
    scratch[256 + tid] = a1;
    scratch[512 + tid] = a2;
    __syncthreads();
    if (tid < 128) {
        an = scratch[tid] + scratch[128 + tid];
        a1 = scratch[256 + tid] + scratch[256 + 128 + tid];
        a2 = scratch[512 + tid] + scratch[512 + 128 + tid];
        const float dm = a1 / an, mean = static_cast<float>(static_cast<_Float16>(biasTab[tid])) + dm; // (the pivot the records were taken around: the bias as a half)
        const float var = fmaxf(a2 / an - dm * dm, 0.0f);
        const float mu = f.gamma[tid] / sqrtf(var + f.eps);
        f.mul[n * 128 + tid] = mu;
        f.shift[n * 128 + tid] = f.beta[tid] - mean * mu; // y = x * mul + shift (instancenorm_fold_kernel's form)
    }
}

// NORM: 0 = no InstanceNorm in front, 1 = its activation is ReLU or none (max(x, lo) on the packed halfs), 2 = any branch-free activation (med3 form)
// FAST: the layer has no batch norm and its activation is none (FAST = 2) or ReLU (FAST = 1): the epilogue is bias + conversion [+ a max on the packed
//       halfs]; 0 = the general bias [-> BN] -> med3 epilogue
// STATS: chain rule F records (+ the in-kernel fold when p.fold.counter)
// D: K steps of weight operands in flight per lane (divides 18)
template <int NORM, bool STATS, int D, int FAST>
__global__ __launch_bounds__(256, 2) void conv2d_widep_kernel(WidePParams p, ActCfg ac, const _Float16* __restrict__ x, const char* __restrict__ wp, const float4* __restrict__ epi,
                                                           _Float16* __restrict__ y) {
    constexpr int NCH = 4, IC = 32 * NCH, OC = 128, S = 18, L = 2 * (D - 1); // L: weight loads younger than the one a step waits for
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l32 = lane & 31, h = lane >> 5;

    int* const syTab = reinterpret_cast<int*>(smem + kLdsSy);
    int* const sxTab = reinterpret_cast<int*>(smem + kLdsSx);
    float* const normTab = reinterpret_cast<float*>(smem + kLdsNorm);
    float* const biasTab = reinterpret_cast<float*>(smem + kLdsBias);

    // ---- once per block
    if (NORM != 0 && tid < 16) normTab[4 * IC + (tid & 7) + (tid >> 3) * IC] = 0.0f; // the zero rows (shift at 4 IC, mul IC floats behind it)
    if (tid < OC) {
        const float4 e4 = epi[tid];
        reinterpret_cast<float4*>(smem + kLdsEpi)[tid] = e4;
        biasTab[tid] = e4.x;
    }
    const unsigned aoff0 = static_cast<unsigned>(((wm * 4 * kTileW + l32) * kQP + h) * 16);   // operand reads: pixel (row 4 wm + t, column l32), slot 2 c8 + h
    const unsigned wlane0 = static_cast<unsigned>(h * 2048 + (wn * 64 + l32) * 16);          // weight operand: [h][oc = 64 wn + 32 u + l32]
    const unsigned G = gridDim.x;
    // Per-lane / per-wave bases the tile loop re-derives from values the compiler cannot see through (an empty asm per iteration): hoisted out of the
    // loop as invariants, the 72 weight-step addresses, the 14 DMA destinations and the epilogue's address arithmetic were 164 spilled scalar and 23-150
    // spilled vector registers (ROCm 7.2; the naive persistent loop of round 4 failed the same way)
    unsigned wlane = wlane0, waveLds = static_cast<unsigned>(wave * 1024);
    unsigned tq = static_cast<unsigned>(tid); // the thread index as the tile loop sees it
    const i4 xRsrc = make_rsrc(x, p.xBytes), yRsrc = make_rsrc(y, p.yBytes);

    // row / column tables of a tile (52 threads, one coordinate each): the source pixel of a staged pixel is separable
    auto resolve_tables = [&](unsigned tile) {
        const unsigned n = tile / p.tilesPerImage, rem = tile - n * p.tilesPerImage, ty = rem / p.tilesX, tx = rem - ty * p.tilesX;
        if (tq >= 192u && tq < 192u + kTileW) {
            const int c = static_cast<int>(tq) - 192;
            int sx = resolve_nobranch(static_cast<int>(tx << 5) - p.padx + c, p.W, p.padMode);
            if (p.preMode) { // a pixel of the (virtual) padded image -> the source pixel the Pad layer would have copied
                const int px = resolve_nobranch(sx - p.preX, p.srcW << p.preShift, p.preMode);
                sx = sx < 0 ? -1 : (px < 0 ? -1 : px >> p.preShift);
            }
            sxTab[c] = sx;
        } else if (tq >= 128u && tq < 128u + kTileH) {
            const int rr = static_cast<int>(tq) - 128;
            int sy = resolve_nobranch(static_cast<int>(ty << 3) - p.pady + rr, p.H, p.padMode);
            if (p.preMode) {
                const int py = resolve_nobranch(sy - p.preY, p.srcH << p.preShift, p.preMode);
                sy = sy < 0 ? -1 : (py < 0 ? -1 : py >> p.preShift);
            }
            syTab[rr] = sy < 0 ? -1 : (static_cast<int>(n) * p.srcH + sy) * p.srcW;
        }
    };
    // staging map of the tile whose tables are published: byte offset of element r's 16 bytes in x (chunk 0), ~0u = zeros (padding, pad slot, past the tile)
    unsigned gofs[kR], gofsN[kR]; // this tile's map / the next tile's (built during chunk 2, used by chunk 3's copies; moved over at the tile's end)
    auto build_map = [&]() {
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const unsigned e = tq + 256u * r;
            const unsigned pix = e / kQP, ql = e - pix * kQP, rr = pix / kTileW, c = pix - rr * kTileW;
            // (rr <= 10 < 32 and c < 34 < 40 for every e < 1792: the look-ups stay inside the tables, no branch around them)
            const int rowPix = syTab[rr], sx = sxTab[c];
            const bool ok = e < static_cast<unsigned>(kTotal) && rowPix >= 0 && sx >= 0 && ql < 4u;
            gofs[r] = ok ? static_cast<unsigned>(rowPix + sx) * (IC * 2) + ql * 16 : kOutOfRange;
        }
    };
    auto map_elem = [&](int r) -> unsigned {
        const unsigned e = tq + 256u * r;
        const unsigned pix = e / kQP, ql = e - pix * kQP, rr = pix / kTileW, c = pix - rr * kTileW;
        const int rowPix = syTab[rr], sx = sxTab[c];
        const bool ok = e < static_cast<unsigned>(kTotal) && rowPix >= 0 && sx >= 0 && ql < 4u;
        return ok ? static_cast<unsigned>(rowPix + sx) * (IC * 2) + ql * 16 : kOutOfRange;
    };
    // one slot of the fix-up in two halves (a K step apart: the LDS round trip hides under that step's MFMAs): read the slot and its table rows ...
    f4 fxv, fxs, fxm;
    const float* fxtb;
    auto fix_read = [&](int bufOfs, int ic0, int slot, int r, unsigned g) {
        // (a padding slot -- zeros from the copy -- looks up the zero rows behind the two table slots: 0 * 0 + 0 stays zero through every activation of the family)
        const float* const tb = g == kOutOfRange ? normTab + 4 * IC : normTab + slot * 2 * IC + ic0 + 8 * ((g >> 4) & 3u);
        fxv = *reinterpret_cast<const f4*>(smem + bufOfs + (tq + 256u * r) * 16);
        fxs = *reinterpret_cast<const f4*>(tb);
        fxm = *reinterpret_cast<const f4*>(tb + IC);
        fxtb = tb;
    };
    auto fix_rows1 = [&]() { // the table rows of the slot's second half, into the registers the first half has just used
        fxs = *reinterpret_cast<const f4*>(fxtb + 4);
        fxm = *reinterpret_cast<const f4*>(fxtb + IC + 4);
    };
    // ... normalise (two halves, one per pair of MFMAs: the ~12 vector instructions of a half fit the issue slots two MFMAs leave), activate, write back
    // (padding stays zero)
    typedef _Float16 h2x __attribute__((ext_vector_type(2)));
    h2x fxo[4];
    auto fix_half = [&](int half) {
        const h8 hv = *reinterpret_cast<const h8*>(&fxv);
#pragma unroll
        for (int k = 4 * half; k < 4 * half + 4; ++k) {
            const float f = fmaf(static_cast<float>(hv[k]), fxm[k & 3], fxs[k & 3]);
            fxo[k >> 1][k & 1] = NORM == 2 ? static_cast<_Float16>(__builtin_amdgcn_fmed3f(fmaxf(f, f * p.normAc.alpha), p.normAc.lo, p.normAc.hi)) : static_cast<_Float16>(f);
        }
        if (NORM == 1) { // max(x, lo) on the rounded halfs, two per instruction (rounding is monotonic and keeps zero: the same bits as max-then-round)
            const _Float16 lo = static_cast<_Float16>(p.normAc.lo); // 0 (ReLU) or -inf (none: the max is the identity)
#pragma unroll
            for (int k2 = 2 * half; k2 < 2 * half + 2; ++k2) fxo[k2] = __builtin_elementwise_max(fxo[k2], h2x{lo, lo});
        }
    };
    auto fix_write = [&](int bufOfs, int r) {
        f4 res;
#pragma unroll
        for (int k = 0; k < 4; ++k) res[k] = __builtin_bit_cast(float, fxo[k]);
        *reinterpret_cast<f4*>(smem + bufOfs + (tq + 256u * r) * 16) = res;
    };
    auto stage_dma = [&](int bufOfs, int ic0) {
#pragma unroll
        for (int r = 0; r < kR; ++r) lds_dma16_buf(xRsrc, gofs[r], static_cast<unsigned>(ic0 * 2), static_cast<unsigned>(bufOfs + 4096 * r) + waveLds);
    };
    // graph rule I: every thread normalises the 16-byte slots ITS lanes copied, between the copies' wait and the barrier that publishes the chunk
    auto norm_fixup = [&](int bufOfs, int ic0, int slot) {
        if (NORM == 0) return;
        const float* const tb0 = normTab + slot * 2 * IC + ic0;
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            // (no branch around a slot of padding: its zeros are normalised like everything else and written back as zeros by the select below -- seven
            // short branches per chunk cost more than the 20 instructions they skip on the few border tiles)
            char* const sp = smem + bufOfs + (tq + 256u * r) * 16;
            const float* const tb = tb0 + 8 * ((gofs[r] >> 4) & 3u);
            const h8 hv = *reinterpret_cast<const h8*>(sp);
            h8 ov;
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const f4 sh = *reinterpret_cast<const f4*>(tb + 4 * q4), mu = *reinterpret_cast<const f4*>(tb + IC + 4 * q4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float f = fmaf(static_cast<float>(hv[4 * q4 + k]), mu[k], sh[k]); // (v_fma_mix: the conversions ride the fma)
                    if (NORM == 2) ov[4 * q4 + k] = static_cast<_Float16>(__builtin_amdgcn_fmed3f(fmaxf(f, f * p.normAc.alpha), p.normAc.lo, p.normAc.hi));
                    else ov[4 * q4 + k] = static_cast<_Float16>(f);
                }
            }
            if (NORM == 1) { // ReLU on the rounded halfs, two per instruction (rounding is monotonic and keeps zero: the same bits as max-then-round)
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    h2 v2 = {ov[2 * k2], ov[2 * k2 + 1]};
                    v2 = __builtin_elementwise_max(v2, h2{static_cast<_Float16>(p.normAc.lo), static_cast<_Float16>(p.normAc.lo)}); // lo: 0 (ReLU) or -inf (none)
                    ov[2 * k2] = v2[0];
                    ov[2 * k2 + 1] = v2[1];
                }
            }
            const bool pad = gofs[r] == kOutOfRange;
            f4 res = *reinterpret_cast<const f4*>(&ov);
#pragma unroll
            for (int k = 0; k < 4; ++k) res[k] = pad ? 0.0f : res[k];
            *reinterpret_cast<f4*>(sp) = res;
        }
    };
    auto load_norm_tab = [&](unsigned n, int slot, float& v) { // thread t < IC: shift[n][t]; t >= IC: mul[n][t - IC]  (IC = 128: one value per thread)
        if (NORM == 0) return;
        const float* src = tq < static_cast<unsigned>(IC) ? p.normShift + n * IC + tq : p.normMul + n * IC + (tq - IC);
        asm volatile("global_load_dword %0, %1, off" : "=&v"(v) : "v"(src));
        (void) slot;
    };

    // ---- first tile: tables, map, chunk 0, weight ring
    // Which run of tiles.  Workgroups go to the eight XCDs round-robin by index: runs handed out by blockIdx.x put vertically neighbouring tiles -- one run
    // apart or two -- into different L2s, and every halo row was fetched from the memory side twice (PMC: 1.5x the algorithmic bytes).  So: segment x of
    // the tile order (one contiguous eighth: two images of a 16-image batch) belongs to XCD x, physical block x + 8 j takes run j of it.  The tiles / G
    // remainder stays with the first-dispatched blocks (physical index < tilesRem), i.e. spread over all XCDs: the first ex(x) runs of segment x are one
    // tile longer.  With G not a multiple of 8 there is one segment and this is the plain scheme (runs by blockIdx.x).
#ifdef SNNHIP_WIDEP_NO_XCD_RUNS // (experiment builds)
    const unsigned nSeg = 1u;
#else
    const unsigned nSeg = (G & 7u) == 0 ? 8u : 1u;
#endif
    const unsigned segRuns = G / nSeg;
    auto seg_extras = [&](unsigned x) -> unsigned { // runs of segment x with tilesBase + 1 tiles
        if (nSeg == 1u) return p.tilesRem;
        return p.tilesRem > x ? min(segRuns, (p.tilesRem - x + 7u) >> 3) : 0u;
    };
    auto run_of_tile = [&](unsigned t) -> unsigned { // the (logical) run that holds tile t
        unsigned x = 0, S = 0;
        for (; x + 1 < nSeg; ++x) {
            const unsigned nextS = S + segRuns * p.tilesBase + seg_extras(x);
            if (t < nextS) break;
            S = nextS;
        }
        const unsigned ex = seg_extras(x), tt = t - S, cut = ex * (p.tilesBase + 1);
        return x * segRuns + (tt < cut ? tt / (p.tilesBase + 1) : ex + (tt - cut) / p.tilesBase);
    };
    const unsigned segX = nSeg == 1u ? 0u : blockIdx.x & 7u, segJ = nSeg == 1u ? blockIdx.x : blockIdx.x >> 3;
    const unsigned bid = segX * segRuns + segJ; // logical run index: the order of the runs in the tile sequence
    unsigned tile = segX * segRuns * p.tilesBase + segJ * p.tilesBase + min(segJ, seg_extras(segX));
    for (unsigned x = 0; x < segX; ++x) tile += seg_extras(x);
    const unsigned runLen = p.tilesBase + (segJ < seg_extras(segX) ? 1u : 0u);
    const unsigned tileEnd = tile + runLen;
    waveLds = __builtin_amdgcn_readfirstlane(waveLds);
    resolve_tables(tile);
    {
        float nv = 0.0f;
        load_norm_tab(tile / p.tilesPerImage, 0, nv);
        if (NORM != 0) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(nv));
            normTab[tid] = nv; // slot 0: [shift[IC] | mul[IC]]
        }
    }
    lds_barrier();
    build_map();
#pragma unroll
    for (int r = 0; r < kR; ++r) gofsN[r] = gofs[r];
    stage_dma(kLdsBuf0, 0);
    f4 bq[D][2];
#pragma unroll
    for (int d = 0; d < D; ++d) gload16x2(wp, wlane + d * kStepBytes, bq[d][0], bq[d][1]);
    vm_wait<0>();
#pragma unroll
    for (int d = 0; d < D; ++d) vm_wait_tie<0>(bq[d][0], bq[d][1]);
    norm_fixup(kLdsBuf0, 0, 0);
    lds_barrier();

    f32x16 acc[4][2];
    f4 a[4];
    int it = 0;            // tiles done by this block (parity = the norm table slot of the current tile)
    float stS1 = 0.0f, stS2 = 0.0f, stCnt = 0.0f; // rule F: this wave's sums for channel 64 wn + lane over the block's tiles of the current image, and their pixel count

#ifdef SNNHIP_WIDEP_TRACE
    unsigned long long wpAcc[8] = {};
    int wpTiles = 0;
    const unsigned long long wpStart = __builtin_amdgcn_s_memrealtime(); // 100 MHz
#endif
    WP_T0();
    for (;;) {
        asm volatile("" : "+v"(wlane), "+v"(tq), "+s"(waveLds)); // (see above: nothing derived from these is a loop invariant)
        const unsigned n = tile / p.tilesPerImage, rem = tile - n * p.tilesPerImage, ty = rem / p.tilesX, tx = rem - ty * p.tilesX;
        const unsigned next = tile + 1;
        const bool hasNext = next < tileEnd;
        const unsigned ntile = hasNext ? next : tile; // (the last tile prefetches itself: the number of copies in the queue must not depend on the tile)
        const int slot = it & 1;
        float nv = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

        // The two blocks of a CU (b and b + G / 2: workgroups fill every CU's first slot before any second one) take turns at the higher wave priority,
        // tile by tile.  Measured, not derived: -1.2 ... -1.6 % on the layer in ABAB runs on two boxes (round 5); the same priority for the whole K loop,
        // for the block dispatched second, or for the wave in its epilogue changed nothing
#ifndef SNNHIP_WIDEP_NO_PRIO_ALT // (experiment builds switch it off)
        if ((it + (blockIdx.x >= (G >> 1) ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(0);
#endif
        // (the four chunks as four calls of one body with a compile-time c instead of a `#pragma unroll` loop: the same code with 20-30 fewer scalar
        // registers spilled to vector lanes -- v_readlane / v_writelane are vector instructions inside the K-steps)
        auto chunk = [&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            const int curOfs = (c & 1) ? kLdsBuf1 : kLdsBuf0, nxtOfs = (c & 1) ? kLdsBuf0 : kLdsBuf1;
            const int nIc0 = c == NCH - 1 ? 0 : (c + 1) * 32, nSlot = c == NCH - 1 ? (slot ^ 1) : slot; // the chunk being staged: the next one (chunk 3: the next tile's first)
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const f4*>(smem + curOfs + aoff0 + t * (kTileW * kQP * 16));
#pragma unroll
            for (int s = 0; s < S; ++s) {
                // KSTEP.  Everything that is not a multiplication rides INSIDE the steps, under the wave's own MFMAs (beside the partner wave's MFMA stream a
                // vector instruction of a non-multiplying phase got an issue slot every ~17 cycles: phase trace): the staged chunk's copy r at
                // step r, the fix-up of its slot r at steps 8 + r / 9 + r, the next tile's tables at chunk 1 step 0 and its map element r at chunk 2 step r.
                // Vector-memory queue at step s: [operand wait] [2 weight loads for step s + D] [copy s, s < 7] [chunk 1, step 0: table load, ticket].
                // The operand of step s was requested at step s - D; younger than it: L = 2 (D - 1) weight loads, the copies of steps s - D .. s - 1 that
                // exist (0 .. 6), the table load when it lies between (thread 0's atomic is left out: a count that is too SMALL only waits for one load
                // more).  Chunk 0, s < D: waited for before the tile loop's back-edge (first tile: by the prologue).
                f4 b0 = bq[s % D][0], b1 = bq[s % D][1];
                if (c >= 1 || s >= D) {
                    const int lo = s - D > 0 ? s - D : 0, hi = s - 1 < kR - 1 ? s - 1 : kR - 1;
                    const int n = L + (hi >= lo ? hi - lo + 1 : 0) + ((c == 1 && NORM != 0 && s >= 1 && s <= D) ? 1 : 0);
                    if (n == L) vm_wait_tie<L>(b0, b1);
                    else if (n == L + 1) vm_wait_tie<L + 1>(b0, b1);
                    else if (n == L + 2) vm_wait_tie<L + 2>(b0, b1);
                    else if (n == L + 3) vm_wait_tie<L + 3>(b0, b1);
                    else vm_wait_tie<L + 4>(b0, b1);
                    static_assert(D == 3, "the chain above covers D = 3 (at most 3 copies + 1 table load between)");
                }
                {
                    const int g2 = (c * S + s + D) % (NCH * S); // the ring wraps into the next tile's first steps: same weights
                    gload16x2(wp, wlane + g2 * kStepBytes, bq[s % D][0], bq[s % D][1]);
                }
                if (s < kR) lds_dma16_buf(xRsrc, c == NCH - 1 ? gofsN[s] : gofs[s], static_cast<unsigned>(nIc0 * 2), static_cast<unsigned>(nxtOfs + 4096 * s) + waveLds);
                if (c == 1 && s == 0) {
                    resolve_tables(ntile);                            // (read by chunk 2's map elements; the previous tile's were last read in ITS chunk 2)
                    load_norm_tab(ntile / p.tilesPerImage, slot ^ 1, nv);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int tap = (s + 1) / 2;
                const int dl = (((tap / 3) * kTileW + (tap % 3)) * kQP + ((s + 1) % 2) * 2) * 16; // compile-time: an immediate offset of the ds_read
                const int fr = s - 8;                                  // fix-up slot read in this step (its copy was issued 8 steps ago and is older than every
                                                                       // weight load the operand wait above left in flight: it has landed)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&b0), *reinterpret_cast<const h8*>(&a[t]), acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&b1), *reinterpret_cast<const h8*>(&a[t]), acc[t][1], 0, 0, 0);
                    if (s + 1 < S) a[t] = *reinterpret_cast<const f4*>(smem + curOfs + aoff0 + t * (kTileW * kQP * 16) + dl);
                    // fix-up, slot fr: read at t = 1, first half + the second half's table rows at t = 3; second half + write-back at t = 0 of the NEXT step
                    if (NORM != 0 && t == 0 && fr - 1 >= 0 && fr - 1 < kR) {
                        fix_half(1);
                        fix_write(nxtOfs, fr - 1);
                    }
                    if (NORM != 0 && t == 1 && fr >= 0 && fr < kR) fix_read(nxtOfs, nIc0, nSlot, fr, c == NCH - 1 ? gofsN[fr] : gofs[fr]);
                    if (NORM != 0 && t == 3 && fr >= 0 && fr < kR) {
                        fix_half(0);
                        fix_rows1();
                    }
                    if (c == 2 && t == 1 && s < kR) gofsN[s] = map_elem(s); // (tables of ntile: published by chunk 1's barrier)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- end of the chunk: everything older than the ring's 2 D youngest loads has landed (the copies, the table load, the ticket)
            WP_ADD(1); // K steps
            // (chunk 1: the norm-table value requested at its step 0 goes THROUGH the wait statement, as the weight ring's registers do in vm_wait_tie -- no
            // use of it can be scheduled in front of the wait; tools/audit_vmcnt.py additionally reports any instruction, a copy or a spill included, that
            // touches the register of an outstanding load)
            if (c == 1 && NORM != 0) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(nv) : "n"(2 * D) : "memory");
            else vm_wait<2 * D>();
            WP_ADD(2); // wait for the copies
            if (c == 1) {
                if (NORM != 0) {
                    normTab[(slot ^ 1) * 2 * IC + tq] = nv;
                }
            }
            WP_ADD(3); // table / ticket bookkeeping
            lds_barrier();
            WP_ADD(4); // barrier
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{});
        static_assert(NCH == 4, "four chunk calls");
#pragma unroll
        for (int r = 0; r < kR; ++r) gofs[r] = gofsN[r];

#ifndef SNNHIP_WIDEP_NO_PRIO_ALT
        __builtin_amdgcn_s_setprio(0);
#endif
        // ---- epilogue (wave-private): accumulator layout acc[t][u][4 g + k] = channel 64 wn + 32 u + 8 g + 4 h + k of pixel (row 4 wm + t, column l32)
        {
            const unsigned lane = tq & 63u, l32 = tq & 31u, h = (tq >> 5) & 1u, wm = (tq >> 6) & 1u, wn = tq >> 7; // (shadow the kernel's: re-derived per tile)
            const unsigned oy0 = (ty << 3) + wm * 4, ox0 = tx << 5;
            char* const scr = smem + kLdsScr + waveLds / 1024 * kScrBytes;
            const unsigned scrW = l32 * kScrPitch + h * 8;                    // + (32 u + 8 g) * 2
            const unsigned scrR = (lane >> 3) * kScrPitch + (lane & 7) * 16;  // + 8 j * pitch
            float sA[8], sB[8]; // rule F: sums and squares of (value - bias) of the 8 channels this lane carries to memory (channel 64 wn + 8 (lane & 7) + e)
#pragma unroll
            for (int e = 0; e < 8; ++e) sA[e] = sB[e] = 0.0f;
            const unsigned pxl = lane >> 3;                              // + 8 j: the pixel (column) of this lane's vector j
            const unsigned laneOut = pxl * (OC * 2) + (wn * 64 + (lane & 7) * 8) * 2; // byte offset of the lane's vector inside a tile row's first 8 pixels
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            f4 bias4[2][4];
            if (FAST) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) bias4[u][g] = *reinterpret_cast<const f4*>(biasTab + wn * 64 + 4 * h + 32 * u + 8 * g);
            }
            // rule F pivots: the channel's bias ROUNDED TO HALF (any pivot serves as long as the fold adds the same one back; a half pivot lets the
            // full-tile path subtract on the packed halfs): as floats for the ragged path, as (p, p) pairs for the packed one
            f4 piv0, piv1;
            h2 pivh[8];
            if (STATS) {
                piv0 = *reinterpret_cast<const f4*>(biasTab + wn * 64 + 8 * (lane & 7));
                piv1 = *reinterpret_cast<const f4*>(biasTab + wn * 64 + 8 * (lane & 7) + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const _Float16 q0 = static_cast<_Float16>(piv0[e]), q1 = static_cast<_Float16>(piv1[e]);
                    pivh[e] = h2{q0, q0};
                    pivh[4 + e] = h2{q1, q1};
                    piv0[e] = static_cast<float>(q0);
                    piv1[e] = static_cast<float>(q1);
                }
            }
            // a tile inside the map (all but the last tile row / column): no per-lane tests, the stores' addresses are one constant lane offset + a scalar
            const bool full = oy0 + 4 <= static_cast<unsigned>(p.OH) && ox0 + 32 <= static_cast<unsigned>(p.OW);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (FAST) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            // (sum first, bias second: the order of conv2d_wide_kernel, whose fused-Add form must give the bits of this layer + an Add launch)
                            // as two aligned pairs: v_pk_add_f32 + v_cvt_pk_f16_f32 each.  Written element by element the vectoriser paired elements
                            // (1, 2) and moved them, and their bias, into fresh register pairs: 13 instructions per four values instead of 4
                            typedef float f2 __attribute__((ext_vector_type(2)));
                            const f2 s01 = f2{acc[t][u][4 * g], acc[t][u][4 * g + 1]} + f2{bias4[u][g][0], bias4[u][g][1]};
                            const f2 s23 = f2{acc[t][u][4 * g + 2], acc[t][u][4 * g + 3]} + f2{bias4[u][g][2], bias4[u][g][3]};
                            h2 lo2 = __builtin_convertvector(s01, h2), hi2 = __builtin_convertvector(s23, h2);
                            if (FAST == 1) { // ReLU on the rounded halfs (same bits as max-then-round)
                                const h2 z2 = {static_cast<_Float16>(0.0f), static_cast<_Float16>(0.0f)};
                                lo2 = __builtin_elementwise_max(lo2, z2);
                                hi2 = __builtin_elementwise_max(hi2, z2);
                            }
                            const h4 o = h4{lo2[0], lo2[1], hi2[0], hi2[1]};
                            *reinterpret_cast<h4*>(scr + scrW + (32 * u + 8 * g) * 2) = o;
                        }
                } else {
                    const float4* const etab = reinterpret_cast<const float4*>(smem + kLdsEpi) + wn * 64 + 4 * h;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            h4 o;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float4 e4 = etab[u * 32 + 8 * g + k]; // {bias, bnScale, bnMean, bnBeta}
                                const float v = epi_affine(acc[t][u][4 * g + k], e4, p.useBN);
                                o[k] = static_cast<_Float16>(apply_act<true>(ac, v, 0.0f));
                            }
                            *reinterpret_cast<h4*>(scr + scrW + (32 * u + 8 * g) * 2) = o;
                        }
                }
                // (one wave's LDS instructions execute in order: the reads below see the writes above, and the next row's writes follow these reads)
                f4 pk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = *reinterpret_cast<const f4*>(scr + scrR + j * 8 * kScrPitch);
                const unsigned oy = oy0 + t;
                const unsigned rowB = ((n * p.OH + oy) * p.OW + ox0) * (OC * 2); // (uniform)
                auto stat_add = [&](const f4& v) {
                    const _Float16* ch = reinterpret_cast<const _Float16*>(&v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = static_cast<float>(ch[e]) - (e < 4 ? piv0[e & 3] : piv1[e & 3]);
                        sA[e] += f;
                        sB[e] = fmaf(f, f, sB[e]);
                    }
                };
                // full tiles: two pixels at a time -- v_perm pairs the two pixels' values of a channel, the pivot comes off both with one packed
                // subtraction, v_dot2c adds (d0 + d1) and (d0 d0 + d1 d1) to the fp32 sums: 2 vector instructions per value instead of 4 (convert,
                // subtract, add, fma).  The subtraction rounds to half: exact whenever value and pivot lie within a factor of two of each other
                // (the offset layers), 2^-11 relative to the DEVIATION otherwise
                auto stat_pair = [&](const f4& va, const f4& vb) {
                    const h2 one2 = {static_cast<_Float16>(1.0f), static_cast<_Float16>(1.0f)};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float fa = va[k], fb = vb[k]; // (element copies first: __builtin_bit_cast on a vector ELEMENT expression reads element 0)
                        const unsigned xa = __builtin_bit_cast(unsigned, fa), xb = __builtin_bit_cast(unsigned, fb);
                        const h2 dl = __builtin_bit_cast(h2, __builtin_amdgcn_perm(xb, xa, 0x05040100u)) - pivh[2 * k];     // channel 2k of both pixels
                        const h2 dh = __builtin_bit_cast(h2, __builtin_amdgcn_perm(xb, xa, 0x07060302u)) - pivh[2 * k + 1]; // channel 2k + 1
                        sA[2 * k] = __builtin_amdgcn_fdot2(dl, one2, sA[2 * k], false);
                        sB[2 * k] = __builtin_amdgcn_fdot2(dl, dl, sB[2 * k], false);
                        sA[2 * k + 1] = __builtin_amdgcn_fdot2(dh, one2, sA[2 * k + 1], false);
                        sB[2 * k + 1] = __builtin_amdgcn_fdot2(dh, dh, sB[2 * k + 1], false);
                    }
                };
                if (full) {
                    if (STATS) {
#ifdef SNNHIP_WIDEP_STATS_SCALAR // experiment build: the per-value form (A/B of the packed one)
#pragma unroll
                        for (int j = 0; j < 4; ++j) stat_add(pk[j]);
#else
                        stat_pair(pk[0], pk[1]);
                        stat_pair(pk[2], pk[3]);
#endif
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) store16_buf(yRsrc, laneOut, rowB + j * 8 * (OC * 2), pk[j]);
                } else {
                    const bool rowIn = oy < static_cast<unsigned>(p.OH);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool in = rowIn && ox0 + pxl + 8 * j < static_cast<unsigned>(p.OW);
                        if (STATS && in) stat_add(pk[j]);
                        // unconditional: a pixel outside the map gets an offset the buffer descriptor drops
                        store16_buf(yRsrc, in ? laneOut : kOutOfRange, rowIn ? rowB + j * 8 * (OC * 2) : 0u, pk[j]);
                    }
                }
            }
            if (STATS) {
                // the wave's record: lanes (pixel group pg = lane >> 3, column c8 = lane & 7) -> LDS [pg][2][64], then lane c sums its channel's 8 groups
                float* const red = reinterpret_cast<float*>(scr); // 8 x 128 floats = 4 096 bytes of the wave's 4 608
                {
                    float* const w = red + (lane >> 3) * 128 + (lane & 7) * 8;
                    *reinterpret_cast<f4*>(w) = f4{sA[0], sA[1], sA[2], sA[3]};
                    *reinterpret_cast<f4*>(w + 4) = f4{sA[4], sA[5], sA[6], sA[7]};
                    *reinterpret_cast<f4*>(w + 64) = f4{sB[0], sB[1], sB[2], sB[3]};
                    *reinterpret_cast<f4*>(w + 68) = f4{sB[4], sB[5], sB[6], sB[7]};
                }
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int g8 = 0; g8 < 8; ++g8) {
                    s1 += red[g8 * 128 + lane];
                    s2 += red[g8 * 128 + 64 + lane];
                }
                const int rows = max(0, min(4, p.OH - static_cast<int>(oy0))), cols = max(0, min(32, p.OW - static_cast<int>(ox0)));
                stS1 += s1;
                stS2 += s2;
                stCnt += static_cast<float>(rows * cols);
            }
        }
        WP_ADD(5); // epilogue
#ifdef SNNHIP_WIDEP_TRACE
        ++wpTiles;
#endif
        if (STATS) {
            // the run leaves the image (or ends): the block's record for image n, then the hand-off of norm_fold.h -- records written through (sc1), acknowledged,
            // the image's counter bumped; whoever writes an image's last record adds them all up.  Once or twice per block and launch (a per-TILE record was
            // two written-through stores in front of every tile's weight loads -- vector memory retires in order -- and an atomic round trip per tile)
            const unsigned nNext = hasNext ? next / p.tilesPerImage : ~0u;
            if (nNext != n) {
                const unsigned lane = tq & 63u, wm = (tq >> 6) & 1u, wn = tq >> 7;
                const unsigned t0 = n * p.tilesPerImage, t1 = t0 + p.tilesPerImage - 1; // first / last tile of the image
                const unsigned bFirst = run_of_tile(t0), bLast = run_of_tile(t1);
                float* const rec = p.statRec + (static_cast<size_t>(n) * p.recsMax + (bid - bFirst)) * kRecFloats;
                float* const po = rec + 4 + wm * 256 + wn * 64 + lane;
                asm volatile("global_store_dword %0, %1, off sc1\n\tglobal_store_dword %0, %2, off offset:512 sc1" ::"v"(po), "v"(stS1), "v"(stS2) : "memory");
                if (wn == 0 && lane == 0) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(rec + wm), "v"(stCnt) : "memory");
                stS1 = stS2 = stCnt = 0.0f;
                vm_wait<0>();
                lds_barrier();
                int* const flag = reinterpret_cast<int*>(smem + kLdsScr + 4 * kScrBytes - 16); // (the scratch's last 16 bytes: pad columns, never data)
                if (tid == 0) {
                    unsigned* cnt = p.counter + n;
                    const unsigned prev = atomicAdd(cnt, 1u);
                    const bool last = prev == bLast - bFirst;
                    if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch (a replayed hipGraph)
                    *flag = last;
                }
                __syncthreads();
                const bool last = *flag != 0;
                __syncthreads();
                if (last) widep_stats_fold(*p.fold, p.statRec + static_cast<size_t>(n) * p.recsMax * kRecFloats, static_cast<int>(bLast - bFirst + 1), biasTab,
                                           reinterpret_cast<float*>(smem + kLdsScr), static_cast<int>(n));
                lds_barrier();
            }
        }
        if (!hasNext) break;
        // the ring (steps 0 .. D-1 of the next tile, requested in chunk 3's last D steps) has landed long ago; formally: everything older than the
        // epilogue's stores.  After this statement no register of the loop's back-edge is the target of a load in flight.
#pragma unroll
        for (int d = 0; d < D; ++d) vm_wait_tie<kEpiStores>(bq[d][0], bq[d][1]);
        WP_ADD(6); // ring wait before the back-edge
        tile = next;
        ++it;
    }
#ifdef SNNHIP_WIDEP_TRACE
    if (tid == 0 && p.N == 15) { // census (a 15-image layer is the experiment's marker): which CU, from when to when
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (8 << 6) | (15 << 11)), xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
        printf("wpblk %d xcc %u hw %x t0 %llu t1 %llu tiles %d\n", blockIdx.x, xcc & 7u, hw & 0xffffu, wpStart, (unsigned long long) __builtin_amdgcn_s_memrealtime(), wpTiles);
    }
    if ((blockIdx.x == 7 || blockIdx.x == 300) && (tid == 0 || tid == 192))
        printf("wptrace blk %d tid %d tiles %d (cycles per tile): top %llu ksteps %llu copywait %llu fixup %llu barrier %llu epilogue %llu ringwait %llu\n", blockIdx.x, tid, wpTiles,
               wpAcc[0] / wpTiles, wpAcc[1] / wpTiles, wpAcc[2] / wpTiles, wpAcc[3] / wpTiles, wpAcc[4] / wpTiles, wpAcc[5] / wpTiles, wpAcc[6] / wpTiles);
#endif

}

typedef void (*WidePFn)(WidePParams, ActCfg, const _Float16*, const char*, const float4*, _Float16*);

struct WidePConvPlan : ConvPlanBase {
    WidePParams p;
    ActCfg ac;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    size_t ldsBytes = 0;
    int gridBlocks = 0;
    int normKind = 0;
    int ringD = 3;
    int fastKind = 0; // 2: no batch norm, activation none; 1: ... ReLU; 0: the general epilogue

    WidePFn pick() const {
#define SNNHIP_WP_F(NK, ST) (fastKind == 2 ? conv2d_widep_kernel<NK, ST, 3, 2> : fastKind == 1 ? conv2d_widep_kernel<NK, ST, 3, 1> : conv2d_widep_kernel<NK, ST, 3, 0>)
        const bool st = p.statRec != nullptr;
        switch (normKind) {
        case 0: return st ? SNNHIP_WP_F(0, true) : SNNHIP_WP_F(0, false);
        case 1: return st ? SNNHIP_WP_F(1, true) : SNNHIP_WP_F(1, false);
        default: return nullptr;
        }
#undef SNNHIP_WP_F
    }

    // chain rule F.  The records are per (image, block): nothing the norm's fold launches could read -- statistics only together with the in-kernel fold
    bool enableTileStats() override {
        if (statPart) return true;
        if (snnhip::option("SNNHIP_NO_KERNEL_FOLD")) return false;
        void* buf = nullptr;
        const size_t bytes = static_cast<size_t>(p.N) * p.recsMax * kRecFloats * sizeof(float);
        if (snnhip::dev_malloc(&buf, bytes, "conv2d_widep block statistics") != hipSuccess) return false;
        deviceAllocs.push_back(buf);
        statPart = p.statRec = static_cast<float*>(buf);
        statTilesX = static_cast<int>(p.recsMax); statTilesY = 1; statTH = 0; statTW = 0;
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
        if (!statPart || p.counter) return false;
        void* buf = nullptr;
        const size_t bytes = static_cast<size_t>(p.N) * sizeof(unsigned);
        if (snnhip::dev_malloc(&buf, bytes, "conv2d_widep image counters") != hipSuccess) return false;
        deviceAllocs.push_back(buf);
        if (hipMemset(buf, 0, bytes) != hipSuccess) return false;
        NormFoldArgs f{};
        f.counter = static_cast<unsigned*>(buf);
        f.gamma = t.gamma; f.beta = t.beta; f.shift = t.shift; f.mul = t.mul; f.eps = t.eps;
        void* fbuf = nullptr;
        if (snnhip::dev_malloc(&fbuf, sizeof(f), "conv2d_widep fold arguments") != hipSuccess) return false;
        deviceAllocs.push_back(fbuf);
        if (hipMemcpy(fbuf, &f, sizeof(f), hipMemcpyHostToDevice) != hipSuccess) return false;
        p.fold = static_cast<const NormFoldArgs*>(fbuf);
        p.counter = f.counter;
        desc += "+fold";
        return true;
    }
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "conv2d: expects 1 input, got %d", nIn);
        SNNHIP_REQUIRE(!p.statRec || p.counter, "conv2d_widep: block statistics were switched on without the in-kernel fold (no fold launch reads its records)");
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.srcH && x->w == p.srcW && x->c == 128 && x->dtype == SNNHIP_F16,
                       "conv2d: input dims %dx%dx%dx%d (dtype %d) != plan %dx%dx%dx%d fp16", x->n, x->h, x->w, x->c, x->dtype, p.N, p.srcH, p.srcW, 128);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == 128 && out->dtype == SNNHIP_F16,
                       "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n, out->h, out->w, out->c, p.N, p.OH, p.OW, 128);
        const WidePFn fn = pick();
        SNNHIP_LAUNCH(fn, dim3(static_cast<unsigned>(gridBlocks)), dim3(256), ldsBytes, ctx->stream, p, ac, reinterpret_cast<const _Float16*>(x->data),
                      reinterpret_cast<const char*>(d_w), reinterpret_cast<const float4*>(d_epi), reinterpret_cast<_Float16*>(out->data));
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

} // namespace

// fp16 3x3 stride-1 layers with IC = OC = 128, no fused residual, at least one tile per resident block slot; SNNHIP_WIDE_PERSIST=0 keeps conv2d_wide_kernel
// (A/B runs)
int make_conv2d_widep_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    if (g.dtype != SNNHIP_F16 || g.kh != 3 || g.kw != 3 || g.sh != 1 || g.sw != 1) return SNNHIP_E_UNSUPPORTED;
    if (g.IC != 128 || g.OC != 128 || !act_is_simple(g.act) || g.addAct >= 0) return SNNHIP_E_UNSUPPORTED; // (other activations / a fused residual: conv2d_wide_kernel)
    if (g.normShift && g.normAct != SNNHIP_ACT_RELU && g.normAct != SNNHIP_ACT_NONE) return SNNHIP_E_UNSUPPORTED; // (other norm activations: conv2d_wide_kernel; the med3 form does not fit 256 registers here)
    if (const char* e = snnhip::option("SNNHIP_WIDE_PERSIST"))
        if (atoi(e) == 0) return SNNHIP_E_UNSUPPORTED;
    if (snnhip::option("SNNHIP_NO_KERNEL_FOLD")) return SNNHIP_E_UNSUPPORTED; // (this kernel's statistics records are per block: only its own fold reads them)
    const int srcH = g.preMode ? g.srcH : g.H, srcW = g.preMode ? g.srcW : g.W;
    const double inBytes = 2.0 * g.N * srcH * srcW * g.IC, outBytes = 2.0 * g.N * g.OH * g.OW * g.OC;
    if (inBytes >= 4026531840.0 || outBytes >= 4026531840.0) return SNNHIP_E_UNSUPPORTED; // 32-bit byte offsets below the out-of-range marker (raw buffer descriptors)
    const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
    const int tilesX = up_div(g.OW, 32), tilesY = up_div(g.OH, 8);
    const long tiles = static_cast<long>(g.N) * tilesX * tilesY;
    const char* force = snnhip::option("SNNHIP_CONV");
    const bool forced = force && strcmp(force, "wide") == 0;
    if (!forced && tiles < 3L * cus / 4) return SNNHIP_E_UNSUPPORTED; // (conv2d_wide's own bound: below it the 128-pixel blocks and their split-K fill the chip better)
    if (tiles >= 2147483647L / 4) return SNNHIP_E_UNSUPPORTED;

    WidePParams p{};
    p.N = g.N; p.H = g.H; p.W = g.W; p.OH = g.OH; p.OW = g.OW; p.padx = g.padx; p.pady = g.pady; p.padMode = g.padMode; p.useBN = g.useBN;
    p.tilesX = tilesX; p.tilesY = tilesY; p.tilesPerImage = tilesX * tilesY; p.numTiles = static_cast<unsigned>(tiles);
    p.preMode = g.preMode; p.preX = g.preX; p.preY = g.preY; p.preShift = g.preMode ? g.preShift : 0;
    p.srcH = srcH; p.srcW = srcW;
    p.normShift = g.normShift; p.normMul = g.normMul;
    p.normAc = make_act_cfg(g.normShift ? g.normAct : SNNHIP_ACT_NONE, g.normLeaky);

    auto* plan = new WidePConvPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * 9);
    plan->epi4 = epi4;
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->normKind = !g.normShift ? 0 : ((g.normAct == SNNHIP_ACT_RELU || g.normAct == SNNHIP_ACT_NONE) ? 1 : 2);
    plan->fastKind = g.useBN ? 0 : (g.act == SNNHIP_ACT_NONE ? 2 : g.act == SNNHIP_ACT_RELU ? 1 : 0);
    plan->ringD = 3;
    plan->ldsBytes = static_cast<size_t>(kLdsNorm) + (2 * 2 * 128 + 128 + 8) * sizeof(float); // two table slots + the zero rows a padding slot looks up
    {   // 80 KB of dynamic LDS: every instantiation this plan may pick later (statistics / fold are switched on after creation)
        bool ok = true;
        for (int st = 0; st < 2 && ok; ++st) {
            plan->p.statRec = st ? reinterpret_cast<float*>(plan) : nullptr; // (only pick()'s test of it)
            ok = hipFuncSetAttribute(reinterpret_cast<const void*>(plan->pick()), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(plan->ldsBytes)) == hipSuccess;
        }
        plan->p.statRec = nullptr;
        if (!ok) {
            set_error("conv2d_widep: hipFuncSetAttribute(%zu) failed", plan->ldsBytes);
            delete plan;
            return SNNHIP_E_HIP;
        }
    }
    plan->gridBlocks = static_cast<int>(std::min<long>(tiles, 2L * cus));
    if (const char* e = snnhip::option("SNNHIP_WIDEP_GRID")) { // experiments: resident blocks
        const int v = atoi(e);
        if (v > 0) plan->gridBlocks = static_cast<int>(std::min<long>(tiles, v));
    }
    p.tilesBase = static_cast<unsigned>(tiles / plan->gridBlocks);
    p.tilesRem = static_cast<unsigned>(tiles % plan->gridBlocks);
    p.recsMax = static_cast<unsigned>(p.tilesPerImage / p.tilesBase + 2); // blocks whose run can touch one image

    // weights: Wp[step = (chunk, tap, c8)][h][oc] x 8 halfs, ic = chunk*32 + (c8*2 + h)*8 + j -- conv2d_wide_f16's packing (C8 = 2) -- + one step of zeros (the padding pixels' DMA source)
    const size_t steps = 72;
    std::vector<float> wpk((steps + 1) * 2 * 128 * 4, 0.0f);
    _Float16* wph = reinterpret_cast<_Float16*>(wpk.data());
    for (int chunk = 0; chunk < 4; ++chunk)
        for (int t = 0; t < 9; ++t)
            for (int c8 = 0; c8 < 2; ++c8)
                for (int hh = 0; hh < 2; ++hh)
                    for (int j = 0; j < 8; ++j) {
                        const int ic = chunk * 32 + (c8 * 2 + hh) * 8 + j;
                        const size_t base = (((static_cast<size_t>(chunk) * 9 + t) * 2 + c8) * 2 + hh) * 128;
                        for (int o = 0; o < 128; ++o) wph[(base + o) * 8 + j] = static_cast<_Float16>(w_oihw[(static_cast<size_t>(o) * 128 + ic) * 9 + t]);
                    }
    std::vector<float> epiP(static_cast<size_t>(128) * 4, 0.0f);
    std::memcpy(epiP.data(), epi4.data(), sizeof(float) * 4 * 128);
    int rc = plan->upload(wpk.data(), wpk.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epiP.data(), epiP.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    p.xBytes = static_cast<unsigned>(inBytes);
    p.yBytes = static_cast<unsigned>(outBytes);
    plan->p = p;
    plan->inDims[0] = g.N; plan->inDims[1] = srcH; plan->inDims[2] = srcW; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->dtype = SNNHIP_F16;
    plan->flops = 2.0 * 9 * g.IC * g.OC * static_cast<double>(g.OH) * g.OW * g.N;
    plan->bytes = 2.0 * (static_cast<double>(g.N) * srcH * srcW * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC + static_cast<double>(g.OC) * g.IC * 9);
    char buf[320];
    snprintf(buf, sizeof(buf), "conv2d_mfma_wide_f16_32x32x16 persistent k=3x3 s=1 ic=128 oc=128 tile=8x32px x 128oc (4x2 MFMA tiles per wave) chunk=32 ring=%d blocks=%d lds=%zuB",
             plan->ringD, plan->gridBlocks, plan->ldsBytes);
    plan->desc = buf;
    if (g.preMode) plan->desc += " +pad(" + std::string(g.preMode == SNNHIP_PAD_REFLECT ? "reflect" : g.preMode == SNNHIP_PAD_REPLICATE ? "replicate" : "constant") + ")";
    if (g.preMode && g.preShift) plan->desc += " +upsample(x2)";
    if (g.normShift) plan->desc = "instancenorm(act=" + std::to_string(g.normAct) + ", in LDS behind the DMA) -> " + plan->desc;
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
