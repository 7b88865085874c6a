// conv2d_stem_f16.hip -- fp16 9x9 stride-1 convolution of a channel-thin image (IC <= 4: the RGB stem of the style-transfer graphs, 3 -> 32) on
// v_mfma_f32_32x32x16_f16.
//
// conv2d_mfma_kernel's tap-pair mode spends one 16-channel K step on two taps of a 3-channel pixel (3 of 8 operand halfs carry data), reads one
// LDS operand per MFMA and walks the taps with a rolled loop: 1.48 ms for 16 x 818 x 1378 x 32 outputs (157 TF/s) where the output alone is
// 1.15 GB (~0.3 ms of HBM) -- and 0.55 ms with every memory operation ablated, i.e. bound by instruction issue.  This kernel instead
//   * keeps a pixel as FOUR halfs (RGB0, 8 bytes) in LDS, so one K step = 4 horizontally adjacent taps x 4 channels: a lane half supplies two
//     adjacent pixels = 16 contiguous bytes.  A 9-tap kernel row = 2 full steps + 1 left-over tap; the left-over taps of 4 kernel rows share a
//     step (rows 0-3, 4-7, 8): 21 K steps per output instead of 41, 96 % of the operand slots carry a tap;
//   * holds ALL weights of a 32-channel output block in registers (21 x 16 bytes per lane, the MFMA's A operand), loaded once per wave;
//   * gives a wave 8 output rows x 32 columns (8 accumulator tiles): the operand of input row r and tap group j is read from LDS ONCE and
//     feeds the MFMAs of every output row y with 0 <= r - y <= 8 -- 0.3 LDS operand reads per MFMA instead of 1;
//   * is straight-line code: 168 MFMAs per wave, no loop, no address arithmetic beyond immediate offsets;
//   * writes each output row through a wave-private LDS scratch (8-byte runs in, 32 contiguous bytes per lane out): a wave stores 2 KB
//     contiguous per row.
// The fused Pad in front (ConvGeom::preMode, reflect in the style graphs) resolves in the staging loop.  Bias / BN / activation as everywhere.
#include "epilogue.h"
#include "norm_fold.h"
#include "snnhip_internal.h"

#include <cstdio>
#include <cstring>

namespace snnhip {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

struct StemParams {
    int N, H, W, IC, OC, padx, pady, padMode, useBN, OH, OW;
    int tilesX, tilesY;
    int preMode, preX, preY, srcH, srcW;
    unsigned xBytes; // size of the input tensor (RGB instantiation: its loads go through a raw buffer descriptor, a read past the end returns zeros)
    unsigned yBytes; // size of the output tensor: the bound of the raw buffer descriptor its stores go through (round 6)
    // chain rule F (Conv2D -> InstanceNorm): per (image, block) records {pixels, sum[32], sum of squares[32]} of the STORED values, and in the image's last
    // block the fold into the norm's shift / mul (norm_fold.h); null = off
    float* statRec;
    int statSlots; // records per image: the most blocks whose tile runs can meet one image
    NormFoldArgs fold;
};

constexpr int kK = 9;                 // kernel extent
constexpr int kNR = 4;                // output rows per wave
constexpr int kTH = 4 * kNR, kTW = 32; // block tile: 4 waves stacked in y
constexpr int kInH = kTH + kK - 1, kInW = kTW + kK - 1; // staged halo tile
constexpr int kSteps = 2 * kK + 3;    // 18 full steps + 3 left-over steps
constexpr int kOutPitch = 40;         // halfs per pixel row of the wave's output scratch (80 bytes: 16-byte aligned rows, the 8-byte runs of 16 lanes on distinct banks)


#ifdef SNNHIP_STEM_TRACE // experiment builds (tools/exp_one.sh): one block prints the s_memtime spans of a tile's phases; the launch's span from its blocks' own clocks
__device__ unsigned long long g_stemFirstEntry = ~0ull, g_stemFirstStart = ~0ull, g_stemLastEnd = 0, g_stemSumLife = 0, g_stemLastStart = 0;
__device__ unsigned g_stemDone = 0;
#define STEM_MARK(i) do { if (strace) sst[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STEM_MARK(i) do { } while (0)
#endif
template <bool SIMPLE, bool STATS /* chain rule F: StemParams::statRec */, bool RGB = false /* IC == 3 (round 6): a pixel is ONE 8-byte load, see stage_load */>
__global__ __launch_bounds__(256, 2) void conv2d_stem_kernel(StemParams p, ActCfg ac, const _Float16* __restrict__ x, const float4* __restrict__ wp,
                                                          const float4* __restrict__ epi, _Float16* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[kInH * kInW * 4];
    __shared__ __attribute__((aligned(16))) _Float16 oscr[4][32 * kOutPitch];
    __shared__ float4 etab[32]; // this block's rows of the epilogue table {bias, bnScale, bnMean, bnBeta}
    __shared__ float red[STATS ? 17 * 256 : 1]; // rule F: the block's partial sums on their way into a record, then the fold's
    __shared__ __attribute__((aligned(16))) float btab[32]; // (STATS) the biases alone: the statistics accumulators take the registers bias16 has otherwise
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, h = lane >> 5;
#ifdef SNNHIP_STEM_TRACE
    if (tid == 0) atomicMin(&g_stemFirstEntry, wall_clock64());
#endif

    // ---- weights of this block's 32 output channels: 21 x 16 bytes per lane, in registers for the life of the wave -- the block is persistent
    // (grid = the resident block count) and walks the pixel tiles with stride gridDim.x, so the 21 KB per wave are fetched once per kernel, not
    // once per 32 KB of output
    float4 wa[kSteps];
    {
        const float4* wsrc = wp + static_cast<size_t>(blockIdx.y) * kSteps * 64 + lane;
#pragma unroll
        for (int s = 0; s < kSteps; ++s) wa[s] = wsrc[s * 64];
    }
    if (tid < 32) {
        etab[tid] = epi[blockIdx.y * 32 + tid];
        btab[tid] = etab[tid].x;
    }
    // (round 6) output stores go through a raw buffer descriptor (base, stride 0, size in bytes): a lane whose pixel lies outside the map hands in an offset beyond
    // the size and the hardware drops its store -- every lane issues every store, no branch around them, so the compiler can COUNT them in front of the next
    // tile's prefetched loads (s_waitcnt vmcnt(8) instead of vmcnt(0): the loads' wait no longer includes the stores' round trip to memory)
    typedef int i4v __attribute__((ext_vector_type(4)));
    typedef int i2v __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t xRsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(x), 0, static_cast<int>(p.xBytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t yRsrc = __builtin_amdgcn_make_buffer_rsrc(y, 0, static_cast<int>(p.yBytes), 0x00020000 /* gfx9 raw buffer: DATA_FORMAT 32 */);
    // layers without batch norm whose activation is none or relu (Candy's stem): the lane's 16 biases stay in registers for the life of the
    // (persistent) wave and a value's epilogue is add + max -- the general form (table row from LDS, run-time batch-norm select, mul / max / med3)
    // is eight instructions a value, 512 per wave and tile next to its 84 MFMAs
    const bool fastEpi = SIMPLE && !p.useBN && ac.alpha == 1.0f && ac.hi == __builtin_huge_valf();
    float bias16[STATS ? 1 : 16];
    if (!STATS) {
#pragma unroll
        for (int i = 0; i < 16; ++i) bias16[i] = epi[blockIdx.y * 32 + 8 * (i >> 2) + 4 * h + (i & 3)].x;
    }

    // staging: pixel -> 4 halfs (channels past IC are 0).  The loads of tile i+1 are issued before the MFMAs of tile i and written to LDS after
    // them: no global-load latency on the critical path (a rolled load-store loop waited out 7 round trips per tile)
    constexpr int kR = (kInH * kInW + 255) / 256;
    _Float16 sv[kR][4];
    i2v svd[kR];      // (RGB) the two dwords around the pixel's 6 bytes
    unsigned svOdd = 0; // (RGB) bit r: the pixel starts in the upper half of its first dword
    unsigned svOk = 0; // bit r: staged element r lies inside the (padded) image; the others are written as zeros
    auto stage_load = [&](int mt) {
        const int tx = mt % p.tilesX, ty = (mt / p.tilesX) % p.tilesY, n = mt / (p.tilesX * p.tilesY);
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int e = tid + 256 * r;
            const int rr = e / kInW, c = e - rr * kInW;
#if defined(SNNHIP_STEM_ABL) && (SNNHIP_STEM_ABL & 1) // ablation builds (tools/exp_one.sh; wrong at the borders): no padding / Pad resolution in the staging
            int sy = min(max(ty * kTH - p.pady + rr, 0), p.srcH - 1), sx = min(max(tx * kTW - p.padx + c, 0), p.srcW - 1);
            if (false) {
#else
            int sy = resolve_nobranch(ty * kTH - p.pady + rr, p.H, p.padMode);
            int sx = resolve_nobranch(tx * kTW - p.padx + c, p.W, p.padMode);
            if (p.preMode) {
#endif
                const int py = resolve_nobranch(sy - p.preY, p.srcH, p.preMode), px = resolve_nobranch(sx - p.preX, p.srcW, p.preMode);
                sy = sy < 0 ? -1 : py;
                sx = sx < 0 ? -1 : px;
            }
            const bool ok = e < kInH * kInW && sy >= 0 && sx >= 0;
            const _Float16* src = x + (static_cast<size_t>(n * p.srcH + (ok ? sy : 0)) * p.srcW + (ok ? sx : 0)) * p.IC;
#if !defined(SNNHIP_STEM_ABL) || !(SNNHIP_STEM_ABL & 16)
#ifndef SNNHIP_STEM_BRANCHY_IO
            // (round 6) EVERY lane loads (a pixel outside the image reads pixel (0, 0) and is zeroed by a select): the loads used to sit behind one exec branch
            // each -- 16 per thread and tile -- and with a load that may or may not have been issued the compiler can only wait with vmcnt(0), which at the
            // next tile's head also waited for this tile's output stores to be acknowledged
            if constexpr (RGB) {
                // an RGB pixel is 6 bytes at byte offset 6 * pixel: the dword pair that starts at the 4-byte boundary below it holds all three halfs -- ONE 8-byte load per
                // pixel instead of three 2-byte ones (ablation: the image loads were 80 of the kernel's 423 us, all of it issue: 16 two-byte loads per thread and tile)
                const unsigned byteOfs = static_cast<unsigned>((n * p.srcH + (ok ? sy : 0)) * p.srcW + (ok ? sx : 0)) * 6u;
                svd[r] = __builtin_amdgcn_raw_buffer_load_b64(xRsrc, static_cast<int>(byteOfs & ~3u), 0, 0);
                svOdd = r == 0 ? ((byteOfs >> 1) & 1u) : (svOdd | (((byteOfs >> 1) & 1u) << r));
                svOk = r == 0 ? (ok ? 1u : 0u) : (svOk | (ok ? 1u << r : 0u));
            } else {
                const int c1 = min(1, p.IC - 1), c2 = min(2, p.IC - 1), c3 = min(3, p.IC - 1); // (uniform; channels past IC re-read the last one and are zeroed)
                // (the raw values stay untouched until the LDS write of the next tile's head: a select here would put the loads' wait right behind them)
                sv[r][0] = src[0]; sv[r][1] = src[c1]; sv[r][2] = src[c2]; sv[r][3] = src[c3];
                svOk = r == 0 ? (ok ? 1u : 0u) : (svOk | (ok ? 1u << r : 0u));
            }
            continue;
#endif
#endif
#if defined(SNNHIP_STEM_ABL) && (SNNHIP_STEM_ABL & 16) // ablation: no image loads
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[r][k] = static_cast<_Float16>(ok ? 1.0f : 0.0f);
            asm volatile("" ::"v"(src));
#else
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[r][k] = (ok && k < p.IC) ? src[k] : static_cast<_Float16>(0.0f);
#endif
        }
    };
    // ---- rule F.  A lane of the store loop below always carries the same 8 channels (8 (lane & 3) ..): sums and sums of squares of the values it
    // stores (the rounded halfs: what a statistics sweep over the tensor would read) accumulate in 17 registers across the block's tiles of one image.
    // The block's tiles are a contiguous run (below) that ascends through one or two images: flush_image writes the block's record of image `img` when
    // the run leaves it (and at the end) and, in the block that counts in last of those whose runs meet the image, folds the image's records into the
    // norm's shift / mul.  No pivot: the values are
    // post-activation halfs of modest range and a record covers ~2 k pixels; var = E[x^2] - mean^2 in fp32 keeps five digits.
    float st1[8], st2[8], stN = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) st1[k] = st2[k] = 0.0f;
    auto flush_image = [&](int img) {
        constexpr int T = 256, BN = 32, RECF = 1 + 2 * BN;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[k * T + tid] = st1[k];
            red[(8 + k) * T + tid] = st2[k];
            st1[k] = st2[k] = 0.0f;
        }
        red[16 * T + tid] = stN;
        stN = 0.0f;
        __syncthreads();
        // the blocks whose runs meet image img: first .. last; this block's record is slot blockIdx.x - first of the image's p.statSlots
        const int tpiF = p.tilesX * p.tilesY, chunkF = (tpiF * p.N + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);
        const int firstB = (img * tpiF) / chunkF, lastB = ((img + 1) * tpiF - 1) / chunkF;
        const int BPI = lastB - firstB + 1;
        float* const rec = p.statRec + (static_cast<size_t>(img * gridDim.y + blockIdx.y) * p.statSlots + (blockIdx.x - firstB)) * RECF;
        if (tid < BN) { // channel tid = 8 piece + k: the 64 threads with (t & 3) == piece
            const int piece = tid >> 3, kk = tid & 7;
            float a1 = 0.0f, a2 = 0.0f, an = 0.0f;
            for (int j = 0; j < 64; ++j) {
                a1 += red[kk * T + piece + 4 * j];
                a2 += red[(8 + kk) * T + piece + 4 * j];
                an += red[16 * T + 4 * j]; // (pixels: counted by the lanes of piece 0)
            }
            st_agent(rec + 1 + tid, a1);
            st_agent(rec + 1 + BN + tid, a2);
            if (tid == 0) st_agent(rec, an);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned* cnt = p.fold.counter + img * gridDim.y + blockIdx.y;
            const unsigned prev = atomicAdd(cnt, 1u);
            const bool last = prev + 1u == static_cast<unsigned>(BPI);
            if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            red[0] = last ? 1.0f : 0.0f;
        }
        __syncthreads();
        const bool last = red[0] != 0.0f;
        __syncthreads();
        if (!last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int ch = tid % BN, part = tid / BN; // 8 parts x 4 records in flight
        const float* const r0 = p.statRec + static_cast<size_t>(img * gridDim.y + blockIdx.y) * p.statSlots * RECF;
        float a1 = 0.0f, a2 = 0.0f, an = 0.0f;
        for (int b0 = part; b0 < BPI; b0 += 4 * 8) {
            float t1[4], t2[4], tn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = b0 + 8 * j;
                t1[j] = t2[j] = tn[j] = 0.0f;
                if (b < BPI) {
                    const float* rb = r0 + static_cast<size_t>(b) * RECF;
                    tn[j] = ld_agent(rb);
                    t1[j] = ld_agent(rb + 1 + ch);
                    t2[j] = ld_agent(rb + 1 + BN + ch);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                an += tn[j];
                a1 += t1[j];
                a2 += t2[j];
            }
        }
        red[tid] = an;
        red[T + tid] = a1;
        red[2 * T + tid] = a2;
        __syncthreads();
        if (tid < BN) {
            an = a1 = a2 = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                an += red[j * BN + tid];
                a1 += red[T + j * BN + tid];
                a2 += red[2 * T + j * BN + tid];
            }
            const float mean = a1 / an;
            const float var = fmaxf(a2 / an - mean * mean, 0.0f);
            const int oc = blockIdx.y * BN + tid;
            const float mu = p.fold.gamma[oc] / sqrtf(var + p.fold.eps);
            p.fold.mul[img * p.OC + oc] = mu;
            p.fold.shift[img * p.OC + oc] = p.fold.beta[oc] - mean * mu;
        }
        __syncthreads();
    };

    // Tile order.  Without the statistics: tiles blockIdx.x, + gridDim.x, ... (the blocks resident at any moment work on neighbouring tiles).  With them:
    // a CONTIGUOUS run of tiles per block, so that a block meets one or two images, not all of them -- a flush is a round trip to the coherence point
    // (record stores acknowledged, a counter fetched), ~10 us: striding through all 16 images of a Candy micro-batch cost every block 16 of them
    // (the kernel 360 -> 610 us); with runs it is one or two, and an image's fold reads ~33 records instead of 512.
    const int tpi = p.tilesX * p.tilesY, totalAll = tpi * p.N;
    const int chunk = (totalAll + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);
    const int tstep = STATS ? 1 : static_cast<int>(gridDim.x);
    const int total = STATS ? min(totalAll, (static_cast<int>(blockIdx.x) + 1) * chunk) : totalAll;
    int mt = STATS ? static_cast<int>(blockIdx.x) * chunk : static_cast<int>(blockIdx.x);
    if (mt >= total) return; // (STATS, block-uniform: the last blocks of a grid that does not divide the tiles)
    int statImg = mt / tpi; // the image whose values the accumulators hold
#ifndef SNNHIP_STEM_BRANCHY_IO
    // (round 6) everything requested so far -- the weights, the epilogue table -- has landed before the tile loop is entered: left pending, the compiler's merged
    // wait for them sat in front of the first MFMA of EVERY tile as s_waitcnt vmcnt(18), which on the back-edge path means "all but two of the previous tile's
    // stores acknowledged"
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
#endif
    stage_load(mt);
#ifndef SNNHIP_STEM_BRANCHY_IO
    // (round 6) eight stores that the hardware drops (offset beyond the descriptor), so that the vector-memory queue at the loop head looks the same on the way in as
    // on the back-edge -- [16 loads of the tile][8 stores] -- and the compiler waits for the loads with vmcnt(8 + ...) on BOTH paths; without them the entry path has
    // no younger stores and the merged wait is vmcnt(0)
#pragma unroll
    for (int i = 0; i < 2 * kNR; ++i) __builtin_amdgcn_raw_buffer_store_b128(i4v{0, 0, 0, 0}, yRsrc, static_cast<int>(0xffffff00u + 16u * i), 0, 0); // (distinct offsets: identical ones are merged into one store)
#endif
#ifdef SNNHIP_STEM_TRACE
    const bool strace = blockIdx.x == 300 && blockIdx.y == 0 && (tid == 0 || tid == 192);
    unsigned long long sst[8] = {};
    int stile = 0;
    const unsigned long long sstWall0 = wall_clock64();
    if (tid == 0) atomicMax(&g_stemLastStart, sstWall0);
    int stilesDone = 0;
#endif
    for (;;) {
        STEM_MARK(0);
        if (STATS) // this tile opens the next image: close the one before it
            for (const int nNow = mt / tpi; statImg < nNow; ++statImg) flush_image(statImg);
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int e = tid + 256 * r;
#ifndef SNNHIP_STEM_BRANCHY_IO
            if constexpr (RGB) {
                const bool ok = (svOk >> r) & 1u, odd = (svOdd >> r) & 1u;
                const unsigned d0 = static_cast<unsigned>(svd[r][0]), d1 = static_cast<unsigned>(svd[r][1]);
                const unsigned rg = odd ? __builtin_amdgcn_alignbit(d1, d0, 16) : d0, b0 = odd ? (d1 >> 16) : (d1 & 0xffffu); // {R, G}, {B, 0}
                if (e < kInH * kInW) *reinterpret_cast<i2v*>(tile + e * 4) = i2v{static_cast<int>(ok ? rg : 0u), static_cast<int>(ok ? b0 : 0u)};
            } else {
                const _Float16 z = static_cast<_Float16>(0.0f);
                const bool ok = (svOk >> r) & 1u;
                if (e < kInH * kInW) *reinterpret_cast<h4*>(tile + e * 4) = h4{ok ? sv[r][0] : z, (ok && 1 < p.IC) ? sv[r][1] : z, (ok && 2 < p.IC) ? sv[r][2] : z, (ok && 3 < p.IC) ? sv[r][3] : z};
            }
#else
            if (e < kInH * kInW) *reinterpret_cast<h4*>(tile + e * 4) = h4{sv[r][0], sv[r][1], sv[r][2], sv[r][3]};
#endif
        }
        STEM_MARK(1);
        __syncthreads();
        STEM_MARK(2);
        const int tx = mt % p.tilesX, ty = (mt / p.tilesX) % p.tilesY, n = mt / (p.tilesX * p.tilesY);
        const int ox0 = tx * kTW, oy0 = ty * kTH;
        const int next = mt + tstep;
        if (next < total) stage_load(next);
        STEM_MARK(3);

        f32x16 acc[kNR];
#pragma unroll
        for (int i = 0; i < kNR; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

        // operand base of this lane: pixel (row wave*kNR, column l32 + 2h) of the tile, in halfs
        const _Float16* const tb = tile + ((wave * kNR) * kInW + l32 + 2 * h) * 4;
        // ---- full steps: input row r (relative to the wave's first output row), tap group j = columns 4j .. 4j+3
#pragma unroll
        for (int r = 0; r < kNR + kK - 1; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // 16 contiguous bytes, 8-byte aligned: two ds_read_b64
                const h4 lo = *reinterpret_cast<const h4*>(tb + (r * kInW + 4 * j) * 4);
                const h4 hi = *reinterpret_cast<const h4*>(tb + (r * kInW + 4 * j + 1) * 4);
                const h8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int yy = 0; yy < kNR; ++yy) {
                    const int ky = r - yy;
                    if (ky < 0 || ky >= kK) continue;
                    acc[yy] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&wa[ky * 2 + j]), b, acc[yy], 0, 0, 0);
                }
            }
        // ---- left-over tap (column 8) of 4 kernel rows per step: lane half h supplies input rows r0 + 2h, r0 + 2h + 1
        const _Float16* const tl = tile + ((wave * kNR + 2 * h) * kInW + l32 + 8) * 4;
#pragma unroll
        for (int r0 = 0; r0 < kNR + 4; ++r0) { // groups 0 (kernel rows 0-3) and 1 (rows 4-7): output rows y = r0 and y = r0 - 4
            const h4 lo = *reinterpret_cast<const h4*>(tl + (r0 * kInW) * 4);
            const h4 hi = *reinterpret_cast<const h4*>(tl + ((r0 + 1) * kInW) * 4);
            const h8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            if (r0 < kNR) acc[r0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&wa[2 * kK + 0]), b, acc[r0], 0, 0, 0);
            if (r0 >= 4 && r0 - 4 < kNR) acc[r0 - 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&wa[2 * kK + 1]), b, acc[r0 - 4], 0, 0, 0);
        }
        // group 2 = kernel row 8 alone: only the first slot of lane half 0 carries a tap, the other three are zeros (weights AND operand: a
        // non-finite pixel outside the receptive field must not reach the accumulator as inf * 0)
#pragma unroll
        for (int yy = 0; yy < kNR; ++yy) {
            h4 lo = *reinterpret_cast<const h4*>(tile + ((wave * kNR + yy + 8) * kInW + l32 + 8) * 4);
            const h4 z = {0, 0, 0, 0};
            if (h) lo = z;
            const h8 b = __builtin_shufflevector(lo, z, 0, 1, 2, 3, 4, 5, 6, 7);
            acc[yy] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&wa[2 * kK + 2]), b, acc[yy], 0, 0, 0);
        }

        STEM_MARK(4);
        // ---- epilogue: acc[yy][4g + k] = channel 8g + 4h + k of pixel (row yy, column l32).  Per row: 8-byte runs -> wave scratch -> 16-byte
        // pieces, lane L = piece L % 4 of pixel L / 4 (+ 16): each store instruction of the wave writes 1 KB contiguous (OC == 32: the row's
        // pixels are adjacent in memory)
        _Float16* const sc = oscr[wave];
#pragma unroll
        for (int yy = 0; yy < kNR; ++yy) {
            const int oy = oy0 + wave * kNR + yy;
            if (fastEpi) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    h4 o;
                    if (STATS) {
                        const float4 b4 = *reinterpret_cast<const float4*>(btab + 8 * g + 4 * h);
                        o[0] = static_cast<_Float16>(fmaxf(acc[yy][4 * g + 0] + b4.x, ac.lo));
                        o[1] = static_cast<_Float16>(fmaxf(acc[yy][4 * g + 1] + b4.y, ac.lo));
                        o[2] = static_cast<_Float16>(fmaxf(acc[yy][4 * g + 2] + b4.z, ac.lo));
                        o[3] = static_cast<_Float16>(fmaxf(acc[yy][4 * g + 3] + b4.w, ac.lo));
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = static_cast<_Float16>(fmaxf(acc[yy][4 * g + k] + bias16[4 * g + k], ac.lo));
                    }
#if defined(SNNHIP_STEM_ABL) && (SNNHIP_STEM_ABL & 2)
                    asm volatile("" ::"v"(o)); // (ablation: the converted row is kept alive, not written to the scratch)
#else
                    *reinterpret_cast<h4*>(sc + l32 * kOutPitch + 8 * g + 4 * h) = o;
#endif
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    h4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float v = epi_affine(acc[yy][4 * g + k], etab[8 * g + 4 * h + k], p.useBN);
                        v = SIMPLE ? apply_act<true>(ac, v, 0.0f) : epi_act(ac.act, ac.leaky, v, 0.0f);
                        o[k] = static_cast<_Float16>(v);
                    }
#if defined(SNNHIP_STEM_ABL) && (SNNHIP_STEM_ABL & 2)
                    asm volatile("" ::"v"(o)); // (ablation: the converted row is kept alive, not written to the scratch)
#else
                    *reinterpret_cast<h4*>(sc + l32 * kOutPitch + 8 * g + 4 * h) = o;
#endif
                }
            }
            // (wave-private scratch: the LDS queue of a wave is in order, no barrier)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int opix = 16 * i + (lane >> 2), piece = lane & 3;
#if defined(SNNHIP_STEM_ABL) && (SNNHIP_STEM_ABL & 2) // ablation: no read-back of the row from the scratch (wrong values, same stores)
                const float4 v = make_float4(acc[yy][4 * i], acc[yy][4 * i + 1], acc[yy][4 * i + 2], acc[yy][4 * i + 3]);
#else
                const float4 v = *reinterpret_cast<const float4*>(sc + opix * kOutPitch + 8 * piece);
#endif
                const int ox = ox0 + opix;
                if (oy < p.OH && ox < p.OW) {
#ifdef SNNHIP_STEM_BRANCHY_IO // the form of rounds 2-5 (A/B builds)
                    *reinterpret_cast<float4*>(y + (static_cast<size_t>(n * p.OH + oy) * p.OW + ox) * p.OC + blockIdx.y * 32 + 8 * piece) = v;
#endif
                    if (STATS) {
                        const h8 hv = *reinterpret_cast<const h8*>(&v);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float xv = static_cast<float>(hv[k]);
                            st1[k] += xv;
                            st2[k] = fmaf(xv, xv, st2[k]);
                        }
                        stN += piece == 0 ? 1.0f : 0.0f;
                    }
                }
                // (the store BEHIND the statistics' branch: in front of it the compiler sank a copy into either arm, and a store that may or may not have been issued
                // cannot be counted)
#ifndef SNNHIP_STEM_BRANCHY_IO
                {
                    const bool inside = oy < p.OH && ox < p.OW;
                    const unsigned off = inside ? static_cast<unsigned>(((n * p.OH + oy) * p.OW + ox) * p.OC + blockIdx.y * 32 + 8 * piece) * 2u : 0xfffffff0u; // (yBytes <= 0xfffffaf0: beyond the descriptor's size, the store is dropped)
#if defined(SNNHIP_STEM_ABL) && (SNNHIP_STEM_ABL & 4) // ablation: no output stores
                    asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(off));
#else
                    i4v vi;
                    vi[0] = __float_as_int(v.x); vi[1] = __float_as_int(v.y); vi[2] = __float_as_int(v.z); vi[3] = __float_as_int(v.w);
                    __builtin_amdgcn_raw_buffer_store_b128(vi, yRsrc, static_cast<int>(off), 0, 0);
#endif
                }
#endif
            }
        }
        STEM_MARK(5);
#ifdef SNNHIP_STEM_TRACE
        ++stilesDone;
        if ((next >= total) && tid == 0 && blockIdx.y == 0) { // the launch's span from its blocks' own clocks: first start .. last end, one line from the block that ends last
            const unsigned long long wEnd = wall_clock64();
            atomicMin(&g_stemFirstStart, sstWall0);
            atomicMax(&g_stemLastEnd, wEnd);
            atomicAdd(&g_stemSumLife, wEnd - sstWall0);
            if (atomicAdd(&g_stemDone, 1u) + 1u == gridDim.x) {
                printf("stemlaunch: %u blocks, first kernel entry .. first loop start %.1f us, first loop start .. last end %.1f us, mean block life %.1f us, last block started %.1f us after the first\n", gridDim.x,
                       (g_stemFirstStart - g_stemFirstEntry) * 0.01, (g_stemLastEnd - g_stemFirstStart) * 0.01, g_stemSumLife * 0.01 / gridDim.x, (g_stemLastStart - g_stemFirstStart) * 0.01);
                g_stemFirstEntry = ~0ull; g_stemDone = 0; g_stemFirstStart = ~0ull; g_stemLastEnd = 0; g_stemSumLife = 0; g_stemLastStart = 0;
            }
        }
#endif
        if (next >= total) break;
        mt = next;
        __syncthreads(); // every wave is done with the tile before the next one is written
#ifdef SNNHIP_STEM_TRACE
        if (strace && ++stile >= SNNHIP_STEM_TRACE && stile < SNNHIP_STEM_TRACE + 3)
            printf("stemtrace tid %d tile %d: wait+ldswrite %llu bar %llu stage_issue %llu mfma %llu epilogue %llu bar2 %llu total %llu\n", tid, stile, sst[1] - sst[0], sst[2] - sst[1],
                   sst[3] - sst[2], sst[4] - sst[3], sst[5] - sst[4], __builtin_readcyclecounter() - sst[5], __builtin_readcyclecounter() - sst[0]);
#endif
    }
    if (STATS) {
        __syncthreads();
        flush_image(statImg);
    }
}

struct StemConvPlan : ConvPlanBase {
    StemParams p;
    ActCfg ac;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    dim3 grid;
    bool simple = true;

    // chain rule F: per (image, block) records, offered only together with the in-kernel fold (conv2d_upconv.hip / conv2d_s2march.hip)
    bool enableTileStats() override {
        if (statPart) return true;
        if (snnhip::option("SNNHIP_NO_KERNEL_FOLD")) return false;
        const int tpi = p.tilesX * p.tilesY, chunk = up_div(tpi * p.N, static_cast<int>(grid.x));
        p.statSlots = up_div(tpi, chunk) + 1;
        void* buf = nullptr;
        if (snnhip::dev_malloc(&buf, static_cast<size_t>(p.N) * grid.y * p.statSlots * (1 + 2 * 32) * sizeof(float)) != hipSuccess) return false;
        deviceAllocs.push_back(buf);
        statPart = p.statRec = static_cast<float*>(buf);
        statTilesX = p.statSlots; statTilesY = 1; statTH = 0; statTW = 0;
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
        SNNHIP_REQUIRE(!p.statRec || p.fold.counter, "conv2d_stem: block statistics were switched on without the in-kernel fold (no fold launch reads its records)");
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.srcH && x->w == p.srcW && x->c == p.IC && x->dtype == SNNHIP_F16,
                       "conv2d: input dims %dx%dx%dx%d (dtype %d) != plan %dx%dx%dx%d fp16", x->n, x->h, x->w, x->c, x->dtype, p.N, p.srcH, p.srcW, p.IC);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.OC && out->dtype == SNNHIP_F16,
                       "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n, out->h, out->w, out->c, p.N, p.OH, p.OW, p.OC);
        auto fn = p.statRec ? (simple ? conv2d_stem_kernel<true, true> : conv2d_stem_kernel<false, true>) : (simple ? conv2d_stem_kernel<true, false> : conv2d_stem_kernel<false, false>);
#ifndef SNNHIP_STEM_BRANCHY_IO
        if (p.IC == 3 && p.xBytes) // (the style graphs' RGB stems)
            fn = p.statRec ? (simple ? conv2d_stem_kernel<true, true, true> : conv2d_stem_kernel<false, true, true>) : (simple ? conv2d_stem_kernel<true, false, true> : conv2d_stem_kernel<false, false, true>);
#endif
        SNNHIP_LAUNCH(fn, grid, dim3(256), 0, ctx->stream, p, ac, reinterpret_cast<const _Float16*>(x->data), reinterpret_cast<const float4*>(d_w),
                      reinterpret_cast<const float4*>(d_epi), reinterpret_cast<_Float16*>(out->data));
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

} // namespace

// fp16, 9x9, stride 1, IC <= 4, OC % 32 == 0, no fused upsampling / residual; SNNHIP_CONV=stem forces nothing more (the shape rule is the
// kernel's contract), SNNHIP_CONV_STEM=0 leaves the layer to conv2d_mfma's tap-pair mode
int make_conv2d_stem_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    if (g.dtype != SNNHIP_F16 || g.kh != kK || g.kw != kK || g.sh != 1 || g.sw != 1 || g.IC > 4 || g.OC % 32 != 0) return SNNHIP_E_UNSUPPORTED;
    if (g.addAct >= 0 || (g.preMode && g.preShift) || g.act == SNNHIP_ACT_SILU_QUIRK) return SNNHIP_E_UNSUPPORTED;
    if (g.normShift) return SNNHIP_E_UNSUPPORTED; // graph rule I: not in this kernel
    if (const char* e = snnhip::option("SNNHIP_CONV_STEM"))
        if (atoi(e) == 0) return SNNHIP_E_UNSUPPORTED;
    if (const char* f = snnhip::option("SNNHIP_CONV"))
        if (strcmp(f, "stem") != 0) return SNNHIP_E_UNSUPPORTED; // another kernel is being forced
    const double outCount = static_cast<double>(g.N) * g.OH * g.OW * g.OC;
    if (outCount >= 2147483000.0) return SNNHIP_E_UNSUPPORTED; // (32-bit element index; the stores' byte offset through the buffer descriptor stays below 2^32 - 16)

    StemParams p{};
    p.N = g.N; p.H = g.H; p.W = g.W; p.IC = g.IC; p.OC = g.OC; p.padx = g.padx; p.pady = g.pady; p.padMode = g.padMode; p.useBN = g.useBN;
    p.OH = g.OH; p.OW = g.OW;
    p.tilesX = up_div(g.OW, kTW); p.tilesY = up_div(g.OH, kTH);
    p.preMode = g.preMode; p.preX = g.preX; p.preY = g.preY;
    p.srcH = g.preMode ? g.srcH : g.H;
    p.srcW = g.preMode ? g.srcW : g.W;
    p.yBytes = static_cast<unsigned>(outCount * 2.0);
    {
        const double inBytes = 2.0 * g.N * p.srcH * p.srcW * g.IC;
        // (0: the two-byte loads.  Rounded up to whole dwords: the last pixel's second dword ends two bytes past the tensor -- inside its allocation, which is
        // rounded to 16 bytes -- and a dword that crosses the descriptor's size would read as zero)
        p.xBytes = inBytes < 4294966000.0 ? (static_cast<unsigned>(inBytes) + 3u) & ~3u : 0u;
    }

    auto* plan = new StemConvPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * kK * kK);
    plan->epi4 = epi4;
    plan->p = p;
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->simple = act_is_simple(g.act);
    {
        const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
        const long tiles = static_cast<long>(p.tilesX) * p.tilesY * g.N;
        plan->grid = dim3(static_cast<unsigned>(std::min<long>(tiles, 2L * cus)), g.OC / 32, 1); // persistent: two blocks per CU
    }

    // weights: Wp[oc block][step][lane = 32 hh + o] x 8 halfs.  Full step s = 2 ky + j: lane half hh holds taps (ky, 4j + 2hh) and (ky, 4j + 2hh + 1),
    // 4 channels each.  Left-over step 18 + q: taps (4q + 2hh, 8) and (4q + 2hh + 1, 8); kernel rows past 8 are zeros.
    const int ocb = g.OC / 32;
    std::vector<float> wpk(static_cast<size_t>(ocb) * kSteps * 64 * 4, 0.0f);
    _Float16* wph = reinterpret_cast<_Float16*>(wpk.data());
    auto wv = [&](int o, int ic, int ky, int kx) { return w_oihw[((static_cast<size_t>(o) * g.IC + ic) * kK + ky) * kK + kx]; };
    for (int b = 0; b < ocb; ++b)
        for (int s = 0; s < kSteps; ++s)
            for (int hh = 0; hh < 2; ++hh)
                for (int o = 0; o < 32; ++o)
                    for (int slot = 0; slot < 2; ++slot) {
                        int ky, kx;
                        if (s < 2 * kK) { ky = s / 2; kx = 4 * (s % 2) + 2 * hh + slot; }
                        else { ky = 4 * (s - 2 * kK) + 2 * hh + slot; kx = 8; }
                        if (ky >= kK) continue;
                        for (int ic = 0; ic < g.IC; ++ic)
                            wph[((static_cast<size_t>(b) * kSteps + s) * 64 + 32 * hh + o) * 8 + slot * 4 + ic] = static_cast<_Float16>(wv(b * 32 + o, ic, ky, kx));
                    }
    int rc = plan->upload(wpk.data(), wpk.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epi4.data(), epi4.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = g.N; plan->inDims[1] = p.srcH; plan->inDims[2] = p.srcW; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->dtype = SNNHIP_F16;
    plan->flops = 2.0 * kK * kK * g.IC * g.OC * static_cast<double>(g.OH) * g.OW * g.N;
    plan->bytes = 2.0 * (static_cast<double>(g.N) * p.srcH * p.srcW * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC + static_cast<double>(g.OC) * g.IC * kK * kK);
    char buf[256];
    snprintf(buf, sizeof(buf), "conv2d_mfma_stem_f16_32x32x16 k=9x9 s=1 ic=%d oc=%d tile=%dx%dpx x 32oc (4 taps x 4 channels per K step, 21 steps, weights in registers)", g.IC,
             g.OC, kTH, kTW);
    plan->desc = buf;
    if (g.preMode) plan->desc += " +pad(" + std::string(g.preMode == SNNHIP_PAD_REFLECT ? "reflect" : g.preMode == SNNHIP_PAD_REPLICATE ? "replicate" : "constant") + ")";
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
