// conv2d_wide_f16.hip -- fp16 3x3 stride-1 convolution for large feature maps (the bodies of the style-transfer / U-Net graphs): the implicit
// GEMM of conv2d_mfma.hip with a 4 x NT register block of v_mfma_f32_32x32x16_f16 tiles per wave (128 pixels x 64 / 32 output channels)
// instead of 2 x 2.
//
// Why a second kernel (measured on Candy 720p fp16, tools/gpu_candy_abl.sh): the 128-pixel blocks of conv2d_mfma_kernel reach 23 % of the
// fp16 matrix peak on the 128 -> 128 layers and 57 % with every memory operation ablated.  One 32x32x16 MFMA consumes 2 KB of operands in 32
// cycles; a CU's four SIMDs would need 256 B/clk with no reuse, LDS delivers 128 B/clk and the vector L1 64 B/clk.  With a 2 x 2 register
// block the activation operands cost 64 B/clk of LDS reads and the weights (streamed per lane from L2) 64 B/clk of L1 -- the L1 is saturated
// at HALF the matrix rate before the halo-tile staging asks for anything.  A 4 x 2 block halves the weight stream (32 B/clk) at the same LDS
// rate, a 256-pixel block tile halves the weight traffic per output once more, and the unrolled K loop spends 8 MFMAs (256 cycles) per LDS /
// L1 round trip instead of 4.
//
//   * block = 256 threads = WM x WN waves, pixel tile TH x 32 (TH = 8 or 16) of ONE image, BN = 32 NT WN output channels:
//       WM=2 WN=2 NT=2: 256 px x 128 oc     WM=4 WN=1 NT=2: 512 px x 64 oc     WM=4 WN=1 NT=1: 512 px x 32 oc
//   * halo tile staged through registers into LDS in channel chunks of 16 C8 (double-buffered, one barrier per chunk), XOR-swizzled 16-byte
//     slots as in conv2d_mfma_kernel (lds_off); the fused Pad / UpSampling address path (ConvGeom::preMode / preShift) resolves here;
//   * graph rule I (the InstanceNorm in front folded into this layer): the DMA carries the raw values, every thread then normalises the slots
//     its own lanes copied, in LDS, before the barrier that publishes the chunk (norm_fixup below);
//   * weights packed exactly as for conv2d_mfma_kernel ([chunk][tap][c8][h][OCp] x 8 halfs), streamed per lane with a 3-step ring;
//   * the weights are the MFMA's A operand, the pixels its B operand: a lane ends up with runs of 4 consecutive output channels of its own
//     pixel and the epilogue (bias / BN / activation / fused residual Add) stores them directly, 8 bytes per lane, without an LDS transpose.
#include "conv2d_mfma_kernel.h"
#include "norm_fold.h"

#include <cstring>

#ifndef SNNHIP_WIDE_ABL
#define SNNHIP_WIDE_ABL 0 // ablation builds only (tools/ablate_wide.sh): 1 no weight refills, 2 no LDS operand reads, 4 no activation DMA, 8 no output stores
#endif

#ifdef SNNHIP_WIDE_TRACE // experiment builds (tools/exp_one.sh): one block prints the s_memtime stamps of its phases
#define WIDE_MARK(i) do { if (wtrace) wstamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define WIDE_MARK(i) do { } while (0)
#endif

namespace snnhip {

namespace {

using mfma_detail::f32x16;

struct WideParams {
    int N, H, W, IC, OC, padx, pady, padMode, useBN, OH, OW;
    int THs;             // log2 of the tile height; the tile is (1 << THs) x 32 pixels
    int tileH, tileW;    // staged halo tile
    int tilesX, tilesY;
    int nChunks, OCp, total, bufFloats;
    int epiOfs;          // float offset in LDS of the block's epilogue-table rows
    int zeroOfs;         // float4 offset (from the packed weights) of a block of zeros: source of the padding pixels' DMA
    int preMode, preX, preY, srcH, srcW, preShift;
    unsigned magicW;
    const void* res; // fused residual Add (chain rule E)
    ActCfg ac2;
    float* statPart; // chain rule F: per (image, tile, channel) {mean, M2} of the stored values, [n][ty][tx][2][OC]; null = off
    NormFoldArgs fold; // ... and the last block of an image folds them into the norm's shift / mul (norm_fold.h); fold.counter == null = off
    // graph rule I: the InstanceNorm in front, normAc(x * mul[n][c] + shift[n][c]), applied to the staged values IN LDS (below); null = none
    const float* normShift;
    const float* normMul;
    ActCfg normAc;
    // K-loop token (round 4): the two blocks a CU holds lock-step -- both in their K loops (the matrix pipe shared), then both in prologue / epilogue
    // / stores (the pipe idle): 0.45 busy over the launch.  kTok[physical CU] is a mutual exclusion around the K loop: a block takes it before its
    // first chunk and returns it after its last, so a CU's blocks ALTERNATE (one multiplies at the full pipe rate while the other stages / stores).
    // Purely a performance device: a token shared by blocks of different CUs (an aliased id) only serialises them.  null = off.
    unsigned* kTok;
};

// physical CU of the calling wave, hashed to [0, 2048): XCC id (8) x the SE / SH / CU fields of HW_ID (bits 8..15)
__device__ __forceinline__ unsigned physical_cu_slot() {
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (8 << 6) | (7 << 11));   // HW_REG_HW_ID, 8 bits from bit 8: cu_id, sh_id, se_id (bit 16 up: the workgroup slot)
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)); // HW_REG_XCC_ID, 4 bits
    return ((xcc & 7u) << 8) | (hw & 255u);
}
constexpr int kTokSlots = 2048;

template <int WM, int WN, int NT, int C8, int R, bool SIMPLE, bool RES>
__global__ __launch_bounds__(256, 2) void conv2d_wide_kernel(WideParams p, ActCfg ac, const _Float16* __restrict__ x, const float4* __restrict__ wp,
                                                          const float4* __restrict__ epi, _Float16* __restrict__ y) {
    static_assert(WM * WN == 4, "4 waves per block");
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    constexpr int MT = 4;
    constexpr int PIX = 32 * MT * WM;
    constexpr int BN = 32 * NT * WN;
    constexpr int Q = 2 * C8;  // 16-byte slots per staged pixel
    constexpr int S = 9 * C8;  // K steps (16 channels each) per chunk
    // weight ring depth in K steps (divides S).  One step is MT * NT MFMAs: 256 cycles with NT = 2 -- three steps of lead cover an L2 round trip --
    // but only 128 with NT = 1, where a whole chunk's 9 steps are kept in flight (64 -> 32 @816x1376 b16: 1500 -> 1277 us; no change for NT = 2)
    constexpr int D = NT == 1 ? 9 : 3;
    constexpr int kTileW = 34; // staged tile width: the 32-pixel tile row + the 3x3 halo
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int l32 = lane & 31, h = lane >> 5;

    const int mt = blockIdx.x;
#ifdef SNNHIP_WIDE_TRACE
    const bool wtrace = (blockIdx.x == 777 || blockIdx.x == 5000) && blockIdx.y == 0 && (tid == 0 || tid == 192);
    unsigned long long wstamp[8] = {};
#endif
    WIDE_MARK(0);
    const int tx = mt % p.tilesX, ty = (mt / p.tilesX) % p.tilesY, n = mt / (p.tilesX * p.tilesY);
    const int ox0 = tx << 5, oy0 = ty << p.THs;
    const int ix0 = ox0 - p.padx, iy0 = oy0 - p.pady;

    // ---- staging.  LDS-DMA (lds_dma16): element e = tid + 256 r lands at 16-byte LDS slot e of the buffer (lane-contiguous, the only
    // placement the instruction offers).  A pixel owns Q + 1 slots: Q of data and one of padding, so that the 32 pixels of an operand row sit
    // (Q + 1) * 16 bytes apart -- an odd number of 16-byte bank slots, conflict-free for ds_read_b128 without an XOR swizzle, and every tap /
    // channel-slot displacement is an immediate offset of the ds_read (with the swizzle of lds_off the unrolled loop kept 72 precomputed
    // addresses in registers and spilled).  Pixels outside the image (zero padding), the padding slots and the slots past the tile read a
    // block of zeros behind the packed weights.  No staging registers, no ds_write; the copy for chunk c+1 is issued at the top of chunk c
    // and waited for with a COUNTED vmcnt before the closing barrier (the weight ring's youngest loads stay in flight).
    // The source pixel of a staged pixel is separable -- its row depends on the tile row only, its column on the tile column only -- so the padding
    // / fused Pad / fused UpSampling resolution runs ONCE per tile row and column (52 threads, one coordinate each) into two small LDS tables, and
    // a thread's R elements are two look-ups each (resolved per element it was ~110 VALU instructions x R: as long as a quarter of the wave's MFMAs)
    constexpr int QP = Q + 1;
    float* const epiTab = smem + p.epiOfs; // (behind the staging buffers AND the epilogue's output tile) this block's BN rows of the epilogue table {bias, bnScale, bnMean, bnBeta}
    float* const biasTab = epiTab + 4 * BN; // the biases alone, contiguous: the four of a lane's channel run are one ds_read_b128
    int* const syTab = reinterpret_cast<int*>(biasTab + BN); // [tileH <= 32] pixel index of the source row's first pixel, -1 = outside (zeros)
    int* const sxTab = syTab + 32;                          // [kTileW] source column, -1 = outside
    float* const normTab = reinterpret_cast<float*>(sxTab + 40); // [2][IC] graph rule I: shift, mul of this block's image
    if (p.normShift)
        for (int i = tid; i < p.IC; i += 256) {
            normTab[i] = p.normShift[n * p.IC + i];
            normTab[p.IC + i] = p.normMul[n * p.IC + i];
        }
    if (tid < BN) {
        const float4 e4 = epi[blockIdx.y * BN + tid];
        reinterpret_cast<float4*>(epiTab)[tid] = e4;
        biasTab[tid] = e4.x;
    }
    if (tid >= 192 && tid < 192 + kTileW) {
        const int c = tid - 192;
        int sx = resolve_nobranch(ix0 + c, p.W, p.padMode);
        if (p.preMode) { // a pixel of the (virtual) padded image -> the source pixel the Pad layer would have copied (-1 stays -1: size <= 2^30)
            const int px = resolve_nobranch(sx - p.preX, p.srcW << p.preShift, p.preMode);
            sx = sx < 0 ? -1 : (px < 0 ? -1 : px >> p.preShift);
        }
        sxTab[c] = sx;
    } else if (tid >= 128 && tid < 128 + p.tileH) {
        const int rr = tid - 128;
        int sy = resolve_nobranch(iy0 + rr, p.H, p.padMode);
        if (p.preMode) {
            const int py = resolve_nobranch(sy - p.preY, p.srcH << p.preShift, p.preMode);
            sy = sy < 0 ? -1 : (py < 0 ? -1 : py >> p.preShift);
        }
        syTab[rr] = sy < 0 ? -1 : (n * p.srcH + sy) * p.srcW;
    }
    __syncthreads();
    int gofs[R];
    unsigned qlPack = 0; // channel slot of element r in bits 4r .. 4r+2 (the norm fix-up below)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = tid + 256 * r;
        gofs[r] = -1;
        if (e < p.total) {
            const int pix = e / QP;
            const int ql = e - pix * QP;
            qlPack |= static_cast<unsigned>(ql) << (4 * r);
            const int rr = static_cast<int>(__umulhi(static_cast<unsigned>(pix), p.magicW));
            const int c = pix - rr * p.tileW;
            const int rowPix = syTab[rr], sx = sxTab[c];
            if (rowPix >= 0 && sx >= 0 && ql < Q) gofs[r] = (rowPix + sx) * p.IC + ql * 8;
        }
    }
    const _Float16* const zeros = reinterpret_cast<const _Float16*>(wp + p.zeroOfs);
    auto stage_dma = [&](float* buf, int ic0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const _Float16* src = gofs[r] >= 0 ? x + gofs[r] + ic0 : zeros;
            lds_dma16(src, buf + (wave * 64 + 256 * r) * 4);
        }
    };

    // graph rule I: the DMA cannot touch the values it carries, so the InstanceNorm in front of this layer is applied to the chunk IN LDS: every
    // thread normalises exactly the 16-byte slots ITS lanes of the DMA wrote (no other thread has seen them: the pass sits between the DMA's
    // vmcnt wait and the barrier that publishes the chunk), x -> half(act(x * mul[c] + shift[c])) in fp32 -- the arithmetic and rounding point of
    // the norm's own normalise sweep, so the result is bit-identical to the separate launches.  Padding (zeros block) stays zero.  Costs ~180
    // instructions per thread and chunk next to the chunk's 144 MFMAs (+10 %); the normalise sweep it replaces is a read + write of the tensor.
    auto norm_fixup = [&](float* buf, int ic0) {
        typedef _Float16 h8v __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (gofs[r] < 0) continue;
            float* slot = buf + (tid + 256 * r) * 4;
            const float* tb = normTab + ic0 + 8 * static_cast<int>((qlPack >> (4 * r)) & 7u);
            h8v hv = *reinterpret_cast<const h8v*>(slot);
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const float4 sh = *reinterpret_cast<const float4*>(tb + 4 * q4), mu = *reinterpret_cast<const float4*>(tb + p.IC + 4 * q4);
                const float shv[4] = {sh.x, sh.y, sh.z, sh.w}, muv[4] = {mu.x, mu.y, mu.z, mu.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float f = fmaf(static_cast<float>(hv[4 * q4 + k]), muv[k], shv[k]);
                    // (one branch-free form for every simple activation -- ReLU is alpha 1, lo 0 --: a per-value select between this and fmaxf(f, 0) computed both)
                    hv[4 * q4 + k] = static_cast<_Float16>(__builtin_amdgcn_fmed3f(fmaxf(f, f * p.normAc.alpha), p.normAc.lo, p.normAc.hi));
                }
            }
            *reinterpret_cast<h8v*>(slot) = hv;
        }
    };

    // ---- MFMA operand addressing: lane (l32, h) reads pixel (row wm*MT + t of the tile, column l32), slot 2 c8 + h; aoff = float offset
    int aoff[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) aoff[t] = (((wm * MT + t) * p.tileW + l32) * QP + h) * 4;
    const int n0 = blockIdx.y * BN + wn * (NT * 32);
    const size_t bstep = static_cast<size_t>(2) * p.OCp; // float4 units per K step
    const float4* bptr = wp + (static_cast<size_t>(h) * p.OCp + n0 + l32);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    float4 bq[D][NT];
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int u = 0; u < NT; ++u) bq[d][u] = bptr[u * 32];
        bptr += bstep;
    }

    WIDE_MARK(1);
    stage_dma(smem, 0);
    lds_dma_wait();
    if (p.normShift) norm_fixup(smem, 0);
    unsigned* const tok = p.kTok ? p.kTok + physical_cu_slot() : nullptr;
    if (tok && tid == 0) { // chunk 0 is in LDS: wait for the CU's other block to leave its K loop
        while (__hip_atomic_exchange(tok, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) __builtin_amdgcn_s_sleep(16);
    }
    __syncthreads();
    WIDE_MARK(2);

    float4 a[MT];
    for (int chunk = 0; chunk < p.nChunks; ++chunk) {
        const float* cur = smem + (chunk & 1) * p.bufFloats;
        const bool more = chunk + 1 < p.nChunks;
        if (more && !(SNNHIP_WIDE_ABL & 4)) stage_dma(smem + ((chunk + 1) & 1) * p.bufFloats, (chunk + 1) * 16 * C8);
#pragma unroll
        for (int t = 0; t < MT; ++t) a[t] = *reinterpret_cast<const float4*>(cur + aoff[t]);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float4 b[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) b[u] = bq[s % D][u];
            if (!(SNNHIP_WIDE_ABL & 1)) {
#pragma unroll
                for (int u = 0; u < NT; ++u) bq[s % D][u] = bptr[u * 32];
                bptr += bstep;
            }
            __builtin_amdgcn_sched_barrier(0); // the refill stays D steps ahead of its use
            const int tap = (s + 1) / C8;
            const int dl = (((tap / 3) * kTileW + (tap % 3)) * QP + ((s + 1) % C8) * 2) * 4; // compile-time: an immediate offset of the ds_read
            // pixel-tile major: a[t] is dead after its NT MFMAs and is refilled for the next step at once -- no second operand set in registers
#pragma unroll
            for (int t = 0; t < MT; ++t) {
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&b[u]), *reinterpret_cast<const h8*>(&a[t]), acc[t][u], 0, 0, 0);
                if (s + 1 < S && !(SNNHIP_WIDE_ABL & 2)) a[t] = *reinterpret_cast<const float4*>(cur + aoff[t] + dl);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the DMA of this chunk is older than every weight refill of the chunk; the ring keeps D * NT loads in flight
        if (D * NT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (D * NT == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if (p.normShift && more) norm_fixup(smem + ((chunk + 1) & 1) * p.bufFloats, (chunk + 1) * 16 * C8);
        __syncthreads();
    }

    if (tok && tid == 0) __hip_atomic_store(tok, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    WIDE_MARK(3);
    // ---- epilogue.  The MFMAs ran with the WEIGHTS as the A operand (M = output channels) and the pixels as B (N = 32 pixels of a tile row), so
    // a lane holds, for ITS pixel, runs of four consecutive output channels: acc[t][u][4g + k] = channel n0 + 32u + 8g + 4h + k of pixel
    // (row wm*MT + t, column l32).  bias -> BN -> activation, each run goes to the LDS tile [pixel][BN halfs] as ONE ds_write_b64 (with the
    // pixels in the accumulator rows it took 128 ds_write_b16 per wave and the epilogue ran as long as the block's MFMAs); the tile then leaves
    // as 16-byte vectors, a pixel's BN channels contiguous (whole 128-byte lines -- storing the 8-byte runs directly was measured 10 % slower
    // than this round trip: 32 partial lines per store instruction).  Fused residual Add on the way out, same rounding points as the separate
    // launches; its loads are all issued before the first store (gfx9 counts loads and stores in one vmcnt and retires them out of order with
    // respect to each other: a load behind a store can only be waited for with vmcnt(0), the full write round trip).
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    constexpr int EPITCH = BN + 8; // halfs per pixel row of the LDS tile: 8-byte runs of 16 consecutive pixels land in distinct banks
    _Float16* const otile = reinterpret_cast<_Float16*>(smem);
    const float4* const etab = reinterpret_cast<const float4*>(epiTab) + wn * (NT * 32) + 4 * h;
    // layers without batch norm (all of the style graphs): the lane's 8 NT biases are read BEFORE the first write of the output tile -- the
    // compiler cannot tell the table from the tile (one LDS array) and serialises a read - wait - write round trip per channel run otherwise
    float4 bias4[NT][4];
    if (!p.useBN) {
        const float4* const btab = reinterpret_cast<const float4*>(biasTab + wn * (NT * 32) + 4 * h);
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias4[u][g] = btab[u * 8 + 2 * g];
    }
    // layers without batch norm whose activation is none or relu (every 3x3 layer of the style graphs): bias and clamp, two instructions per value --
    // the general form below (run-time batch-norm select, mul / max / med3 activation) is seven, and with 128 values per lane the epilogue's
    // VALU work was a third of the wave's MFMA time
    const bool fastEpi = SIMPLE && !p.useBN && ac.alpha == 1.0f && ac.hi == __builtin_huge_valf();
    if (fastEpi) {
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float bk[4] = {bias4[u][g].x, bias4[u][g].y, bias4[u][g].z, bias4[u][g].w};
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    h4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = static_cast<_Float16>(fmaxf(acc[t][u][4 * g + k] + bk[k], ac.lo));
                    *reinterpret_cast<h4*>(otile + ((wm * MT + t) * 32 + l32) * EPITCH + wn * (NT * 32) + u * 32 + 8 * g + 4 * h) = o;
                }
            }
    } else
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 e[4];
            if (p.useBN) {
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = etab[u * 32 + 8 * g + k];
            } else {
                e[0].x = bias4[u][g].x; e[1].x = bias4[u][g].y; e[2].x = bias4[u][g].z; e[3].x = bias4[u][g].w;
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                h4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = epi_affine(acc[t][u][4 * g + k], e[k], p.useBN);
                    v = SIMPLE ? apply_act<true>(ac, v, 0.0f) : epi_act(ac.act, ac.leaky, v, 0.0f);
                    o[k] = static_cast<_Float16>(v);
                }
                *reinterpret_cast<h4*>(otile + ((wm * MT + t) * 32 + l32) * EPITCH + wn * (NT * 32) + u * 32 + 8 * g + 4 * h) = o;
            }
        }
    constexpr int VPR = BN / 8;             // 16-byte vectors per pixel of the tile
    constexpr int NV = PIX * VPR / 256;     // ... per thread
    int oofs[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int v = tid + 256 * j;
        const int i = v / VPR, c8 = v - i * VPR;
        const int oy = oy0 + (i >> 5), ox = ox0 + (i & 31);
        oofs[j] = (oy < p.OH && ox < p.OW) ? ((n * p.OH + oy) * p.OW + ox) * p.OC + blockIdx.y * BN + c8 * 8 : -1;
    }
    WIDE_MARK(4);
    float4 rpack[RES ? NV : 1];
    if (RES) {
#pragma unroll
        for (int j = 0; j < NV; ++j) rpack[j] = oofs[j] >= 0 ? *reinterpret_cast<const float4*>(static_cast<const _Float16*>(p.res) + oofs[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // chain rule F (p.statPart, uniform): the InstanceNorm behind this layer needs mean and variance per (image, channel).  A thread's NV vectors are
    // NV pixels of ONE 8-channel column (256 % VPR == 0): sums of (v - pivot) and (v - pivot)^2 of the stored (rounded) values accumulate in
    // registers while the vectors pass through, pivot = the thread's first pixel (the differences stay small: no cancellation in S2 - S1^2 / n)
    WIDE_MARK(5);
    float sPiv[8], sA[8], sB[8];
    int sCnt = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) sPiv[e] = sA[e] = sB[e] = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int v = tid + 256 * j;
        const int i = v / VPR, c8 = v - i * VPR;
        float4 pack = *reinterpret_cast<const float4*>(otile + i * EPITCH + c8 * 8);
        if (!RES && p.statPart && oofs[j] >= 0) {
            const _Float16* ch = reinterpret_cast<const _Float16*>(&pack);
            if (sCnt == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sPiv[e] = static_cast<float>(ch[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = static_cast<float>(ch[e]) - sPiv[e];
                sA[e] += f;
                sB[e] = fmaf(f, f, sB[e]);
            }
            ++sCnt;
        }
        if (RES) {
            const _Float16* ch = reinterpret_cast<const _Float16*>(&pack);
            const _Float16* rh = reinterpret_cast<const _Float16*>(&rpack[RES ? j : 0]);
            _Float16 oh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) oh[e] = static_cast<_Float16>(add_act(p.ac2, true, static_cast<float>(ch[e]) + static_cast<float>(rh[e])));
            pack = *reinterpret_cast<const float4*>(oh);
        }
        if (oofs[j] >= 0 && !((SNNHIP_WIDE_ABL & 8) && p.OC != 12345)) *reinterpret_cast<float4*>(y + oofs[j]) = pack;
    }
    WIDE_MARK(6);
#ifdef SNNHIP_WIDE_TRACE
    if (wtrace && !(!RES && p.statPart))
        printf("widetrace blk %d tid %d: prologue %llu stage0 %llu kloop %llu epi %llu bar %llu stores %llu total %llu\n", blockIdx.x, tid, wstamp[1] - wstamp[0], wstamp[2] - wstamp[1],
               wstamp[3] - wstamp[2], wstamp[4] - wstamp[3], wstamp[5] - wstamp[4], wstamp[6] - wstamp[5], wstamp[6] - wstamp[0]);
#endif
    if (!RES && p.statPart) {
        // thread record (count, mean, M2) per channel -> LDS (the output tile is dead); thread c < BN then merges the PG pixel groups of its channel in a
        // fixed order with the parallel-variance update (n = na + nb, d = mb - ma, m = ma + d nb / n, M2 = M2a + M2b + d^2 na nb / n): deterministic,
        // and as accurate as the two-pass form whatever the mean / deviation ratio of the layer
        constexpr int PG = 256 / VPR;
        float* const sred = smem; // [3][PG][BN]
        __syncthreads();
        {
            const int c8 = tid % VPR, pg = tid / VPR;
            const float cn = static_cast<float>(sCnt), inv = sCnt > 0 ? 1.0f / cn : 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float m1 = sA[e] * inv;
                sred[pg * BN + c8 * 8 + e] = sPiv[e] + m1;
                sred[(PG + pg) * BN + c8 * 8 + e] = fmaxf(sB[e] - sA[e] * m1, 0.0f);
            }
            if (c8 == 0) sred[2 * PG * BN + pg] = cn;
        }
        __syncthreads();
        // two levels, all loads ahead of the dependent arithmetic: thread (channel c, part q) merges 8 pixel groups, thread c < BN the 256 / BN parts
        // (one thread per channel walking all PG = 16..64 groups was a 64-step chain of LDS round trips and divisions: +60 % on the 32-channel blocks)
        constexpr int PARTS = 256 / BN, GP = PG / PARTS;
        float* const sred2 = sred + 2 * PG * BN + PG; // [3][PARTS][BN]
        {
            const int c = tid % BN, q = tid / BN;
            float nb[GP], mb[GP], qb[GP];
#pragma unroll
            for (int j = 0; j < GP; ++j) {
                nb[j] = sred[2 * PG * BN + q * GP + j];
                mb[j] = sred[(q * GP + j) * BN + c];
                qb[j] = sred[(PG + q * GP + j) * BN + c];
            }
            float na = 0.0f, ma = 0.0f, M2 = 0.0f;
#pragma unroll
            for (int j = 0; j < GP; ++j) {
                const float n2 = na + nb[j], d = mb[j] - ma, r = nb[j] * __builtin_amdgcn_rcpf(fmaxf(n2, 1.0f)); // nb = 0 (a group outside the image): no change
                ma = fmaf(d, r, ma);
                M2 += qb[j] + d * d * na * r;
                na = n2;
            }
            sred2[q * BN + c] = na;
            sred2[(PARTS + q) * BN + c] = ma;
            sred2[(2 * PARTS + q) * BN + c] = M2;
        }
        __syncthreads();
        if (tid < BN) {
            float nb[PARTS], mb[PARTS], qb[PARTS];
#pragma unroll
            for (int j = 0; j < PARTS; ++j) {
                nb[j] = sred2[j * BN + tid];
                mb[j] = sred2[(PARTS + j) * BN + tid];
                qb[j] = sred2[(2 * PARTS + j) * BN + tid];
            }
            float na = 0.0f, ma = 0.0f, M2 = 0.0f;
#pragma unroll
            for (int j = 0; j < PARTS; ++j) {
                const float n2 = na + nb[j], d = mb[j] - ma, r = nb[j] * __builtin_amdgcn_rcpf(fmaxf(n2, 1.0f));
                ma = fmaf(d, r, ma);
                M2 += qb[j] + d * d * na * r;
                na = n2;
            }
            float* po = p.statPart + (static_cast<size_t>(n * p.tilesY + ty) * p.tilesX + tx) * 2 * p.OC + blockIdx.y * BN;
            st_agent(po + tid, ma); // (read by another block of this launch when the kernel folds: norm_fold.h)
            st_agent(po + p.OC + tid, M2);
        }
        WIDE_MARK(7);
#ifdef SNNHIP_WIDE_TRACE
        if (wtrace)
            printf("widetrace blk %d tid %d: prologue %llu stage0 %llu kloop %llu epi %llu bar %llu stores %llu stats %llu total %llu\n", blockIdx.x, tid, wstamp[1] - wstamp[0],
                   wstamp[2] - wstamp[1], wstamp[3] - wstamp[2], wstamp[4] - wstamp[3], wstamp[5] - wstamp[4], wstamp[6] - wstamp[5], wstamp[7] - wstamp[6], wstamp[7] - wstamp[0]);
#endif
        if (p.fold.counter) tile_stats_finish<BN>(p.fold, p.statPart, sred2 + 3 * 256, n, p.tilesX, p.tilesY, 1 << p.THs, 32, p.OH, p.OW, p.OC, blockIdx.y * BN);
    }
}

typedef void (*WideFn)(WideParams, ActCfg, const _Float16*, const float4*, const float4*, _Float16*);

struct WideConvPlan : ConvPlanBase {
    WideParams p;
    ActCfg ac;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    size_t ldsBytes = 0;
    dim3 grid;
    WideFn kernel = nullptr;
    bool fusedAdd = false;

    // chain rule F: {mean, M2} of every output tile and channel next to the output (the epilogue's LDS round trip carries them: conv2d_wide_kernel)
    bool enableTileStats() override {
        if (statPart) return true;
        if (fusedAdd) return false; // (rule E: the kernel instantiation with the residual carries no statistics code)
        void* buf = nullptr;
        const size_t bytes = static_cast<size_t>(p.N) * p.tilesY * p.tilesX * 2 * p.OC * sizeof(float);
        if (snnhip::dev_malloc(&buf, bytes) != hipSuccess) return false;
        deviceAllocs.push_back(buf);
        statPart = p.statPart = static_cast<float*>(buf);
        statTilesX = p.tilesX; statTilesY = p.tilesY; statTH = 1 << p.THs; statTW = 32;
        desc += " +tile-stats";
        return true;
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
        SNNHIP_REQUIRE(nIn == (fusedAdd ? 2 : 1), "conv2d: expects %d input(s), got %d", fusedAdd ? 2 : 1, nIn);
        const snnhip_tensor* x = in[0];
        WideParams q = p;
        q.res = nullptr;
        if (fusedAdd) {
            const snnhip_tensor* r = in[1];
            SNNHIP_REQUIRE(r->n == p.N && r->h == p.OH && r->w == p.OW && r->c == p.OC && r->dtype == dtype,
                           "conv2d+add: residual %dx%dx%dx%d (dtype %d) does not match the output %dx%dx%dx%d", r->n, r->h, r->w, r->c, r->dtype, p.N, p.OH,
                           p.OW, p.OC);
            q.res = r->data;
        }
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.srcH && x->w == p.srcW && x->c == p.IC && x->dtype == SNNHIP_F16,
                       "conv2d: input dims %dx%dx%dx%d (dtype %d) != plan %dx%dx%dx%d fp16", x->n, x->h, x->w, x->c, x->dtype, p.N, p.srcH, p.srcW, p.IC);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.OC && out->dtype == SNNHIP_F16,
                       "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n, out->h, out->w, out->c, p.N, p.OH, p.OW, p.OC);
        SNNHIP_LAUNCH(kernel, grid, dim3(256), ldsBytes, ctx->stream, q, ac, reinterpret_cast<const _Float16*>(x->data), reinterpret_cast<const float4*>(d_w),
                           reinterpret_cast<const float4*>(d_epi), reinterpret_cast<_Float16*>(out->data));
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

template <int WM, int WN, int NT, int C8, int R>
WideFn pick_wide(bool simple, bool res) {
    if (res) return simple ? conv2d_wide_kernel<WM, WN, NT, C8, R, true, true> : conv2d_wide_kernel<WM, WN, NT, C8, R, false, true>;
    return simple ? conv2d_wide_kernel<WM, WN, NT, C8, R, true, false> : conv2d_wide_kernel<WM, WN, NT, C8, R, false, false>;
}

} // namespace

// fp16 3x3 stride-1 layers with IC % 16 == 0, OC % 32 == 0 and at least 0.75 blocks of 256 / 512 pixels per CU; SNNHIP_CONV=wide forces it for
// every eligible shape, SNNHIP_CONV_WIDE=0 keeps the 128-pixel kernel
int make_conv2d_wide_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    if (g.dtype != SNNHIP_F16 || g.kh != 3 || g.kw != 3 || g.sh != 1 || g.sw != 1) return SNNHIP_E_UNSUPPORTED;
    if (g.IC % 16 != 0 || g.OC % 32 != 0 || g.act == SNNHIP_ACT_SILU_QUIRK) return SNNHIP_E_UNSUPPORTED;
    if (g.normShift && !act_is_simple(g.normAct)) return SNNHIP_E_UNSUPPORTED; // graph rule I: the branch-free activations only
    const double inCount = static_cast<double>(g.N) * (g.preMode ? g.srcH : g.H) * (g.preMode ? g.srcW : g.W) * g.IC;
    const double outCount = static_cast<double>(g.N) * g.OH * g.OW * g.OC;
    if (inCount >= 2147483647.0 || outCount >= 2147483647.0) return SNNHIP_E_UNSUPPORTED;
    const char* force = snnhip::option("SNNHIP_CONV");
    const bool forced = force && strcmp(force, "wide") == 0;
    // the 128 -> 128 body layers of the style graphs: the persistent form (conv2d_widep_f16.hip), unless it declines or SNNHIP_WIDE_PERSIST=0
    if (make_conv2d_widep_plan(ctx, g, w_oihw, epi4, out) == SNNHIP_OK) return SNNHIP_OK;

    // block shape: 256 px x 128 oc when the channels fill it, else 512 px x 64 / 32 oc
    int WM = 4, NT = 2, BN = 64;
    if (g.OC % 128 == 0) { WM = 2; NT = 2; BN = 128; }
    else if (g.OC % 64 != 0) { NT = 1; BN = 32; }
    if (const char* e = snnhip::option("SNNHIP_WIDE_BN")) { // experiments
        const int v = atoi(e);
        if (v == 128 && g.OC % 128 == 0) { WM = 2; NT = 2; BN = 128; }
        if (v == 64 && g.OC % 64 == 0) { WM = 4; NT = 2; BN = 64; }
        if (v == 32) { WM = 4; NT = 1; BN = 32; }
    }
    const int THs = WM == 2 ? 3 : 4, TH = 1 << THs, TW = 32;
    const int C8 = (WM == 2 && g.IC % 32 == 0) ? 2 : 1;
    const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
    const long tiles = static_cast<long>(g.N) * up_div(g.OH, TH) * up_div(g.OW, TW);
    // measured down to one 183x323 image (230 blocks of 256 pixels on 256 CUs): 24 us here, 35 us with the 128-pixel blocks; below that the
    // narrow blocks (and their split-K) fill the chip better
    if (!forced && tiles * (g.OC / BN) * 4 < 3L * cus) return SNNHIP_E_UNSUPPORTED;

    WideParams p{};
    p.N = g.N; p.H = g.H; p.W = g.W; p.IC = g.IC; p.OC = g.OC; p.padx = g.padx; p.pady = g.pady; p.padMode = g.padMode; p.useBN = g.useBN;
    p.OH = g.OH; p.OW = g.OW; p.THs = THs;
    p.tileH = TH + 2; p.tileW = TW + 2;
    p.tilesX = up_div(g.OW, TW); p.tilesY = up_div(g.OH, TH);
    p.nChunks = g.IC / (16 * C8);
    p.OCp = g.OC;
    p.total = p.tileH * p.tileW * (2 * C8 + 1); // 16-byte slots: Q of data + 1 of padding per pixel
    const int R = up_div(p.total, 256);
    p.bufFloats = R * 256 * 4; // every DMA lane has a slot
    p.zeroOfs = static_cast<int>((static_cast<size_t>(p.nChunks) * 9 * C8 + 3) * 2 * g.OC); // the last of the 4 zero steps behind the weights
    p.preMode = g.preMode; p.preX = g.preX; p.preY = g.preY; p.preShift = g.preMode ? g.preShift : 0;
    p.srcH = g.preMode ? g.srcH : g.H;
    p.srcW = g.preMode ? g.srcW : g.W;
    p.magicW = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(p.tileW) - 1) / static_cast<unsigned>(p.tileW));
    p.res = nullptr;
    p.ac2 = make_act_cfg(g.addAct >= 0 ? g.addAct : SNNHIP_ACT_NONE, g.addLeaky);

    const bool simple = act_is_simple(g.act);
    const bool withRes = g.addAct >= 0;
    if (withRes && !act_is_simple(g.addAct)) return SNNHIP_E_UNSUPPORTED; // (no graph of the zoo has one)
    WideFn fn = nullptr;
    if (WM == 2 && C8 == 2 && R <= 7) fn = pick_wide<2, 2, 2, 2, 7>(simple, withRes);
    if (WM == 2 && C8 == 1 && R <= 4) fn = pick_wide<2, 2, 2, 1, 4>(simple, withRes);
    if (WM == 4 && NT == 2 && R <= 8) fn = pick_wide<4, 1, 2, 1, 8>(simple, withRes);
    if (WM == 4 && NT == 1 && R <= 8) fn = pick_wide<4, 1, 1, 1, 8>(simple, withRes);
    if (!fn) return SNNHIP_E_UNSUPPORTED;
    const int PIX = 128 * WM;
    p.epiOfs = static_cast<int>(std::max(static_cast<size_t>(2) * p.bufFloats * 4, static_cast<size_t>(PIX) * (BN + 8) * 2) / 4);
    const size_t lds = static_cast<size_t>(p.epiOfs) * 4 + static_cast<size_t>(BN) * 20 + (32 + 40) * sizeof(int) + // + the row / column tables of the staging
                       (g.normShift ? static_cast<size_t>(2) * g.IC * sizeof(float) : 0);                                  // + the norm's shift / mul (rule I)
    p.normShift = g.normShift;
    p.normMul = g.normMul;
    p.normAc = make_act_cfg(g.normShift ? g.normAct : SNNHIP_ACT_NONE, g.normLeaky);
    if (lds > 80 * 1024) return SNNHIP_E_UNSUPPORTED;

    auto* plan = new WideConvPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * 9);
    plan->epi4 = epi4;
    plan->p = p;
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->kernel = fn;
#ifdef SNNHIP_WIDE_LDS_PAD_KB // experiment builds (tools/exp_one.sh): extra LDS per block = fewer blocks per CU (what does a block do when it has the CU to itself?)
    plan->ldsBytes = lds + static_cast<size_t>(SNNHIP_WIDE_LDS_PAD_KB) * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(plan->kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(plan->ldsBytes));
#else
    plan->ldsBytes = lds;
#endif
    plan->fusedAdd = g.addAct >= 0;
    if (plan->fusedAdd) plan->numInputs = 2;
    plan->grid = dim3(static_cast<unsigned>(tiles), g.OC / BN, 1);
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
        set_error("conv2d_wide: hipFuncSetAttribute(%zu) failed", lds);
        delete plan;
        return SNNHIP_E_HIP;
    }

    // weights: Wp[chunk][tap][c8][h][OC] x 8 halfs, ic = chunk*16*C8 + (c8*2 + h)*8 + j  (+ 10 zero steps: the ring's read-ahead, D <= 9, and the DMA's block of zeros)
    const size_t steps = static_cast<size_t>(p.nChunks) * 9 * C8;
    std::vector<float> wpk((steps + 10) * 2 * g.OC * 4, 0.0f); // the ring reads up to D = 9 steps past the last one
    _Float16* wph = reinterpret_cast<_Float16*>(wpk.data());
    for (int chunk = 0; chunk < p.nChunks; ++chunk)
        for (int t = 0; t < 9; ++t)
            for (int c8 = 0; c8 < C8; ++c8)
                for (int hh = 0; hh < 2; ++hh)
                    for (int j = 0; j < 8; ++j) {
                        const int ic = chunk * 16 * C8 + (c8 * 2 + hh) * 8 + j;
                        const size_t base = (((static_cast<size_t>(chunk) * 9 + t) * C8 + c8) * 2 + hh) * g.OC;
                        for (int o = 0; o < g.OC; ++o) wph[(base + o) * 8 + j] = static_cast<_Float16>(w_oihw[(static_cast<size_t>(o) * g.IC + ic) * 9 + t]);
                    }
    std::vector<float> epiP(static_cast<size_t>(g.OC) * 4, 0.0f);
    std::memcpy(epiP.data(), epi4.data(), sizeof(float) * 4 * static_cast<size_t>(g.OC));
    int rc = plan->upload(wpk.data(), wpk.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epiP.data(), epiP.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    // K-loop token: where a CU holds exactly two of these blocks and the grid is several rounds of them (SNNHIP_WIDE_TOKEN=0 / 1 forces it off / on)
    {
        const char* tk = snnhip::option("SNNHIP_WIDE_TOKEN");
        const bool twoPerCu = lds > 53 * 1024;
        const bool want = tk ? atoi(tk) != 0 : false;
        if (want && twoPerCu) {
            void* buf = nullptr;
            if (snnhip::dev_malloc(&buf, kTokSlots * sizeof(unsigned)) == hipSuccess) {
                plan->deviceAllocs.push_back(buf);
                if (hipMemset(buf, 0, kTokSlots * sizeof(unsigned)) == hipSuccess) plan->p.kTok = static_cast<unsigned*>(buf);
            }
        }
    }
    plan->inDims[0] = g.N; plan->inDims[1] = p.srcH; plan->inDims[2] = p.srcW; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->dtype = SNNHIP_F16;
    plan->flops = 2.0 * 9 * g.IC * g.OC * static_cast<double>(g.OH) * g.OW * g.N;
    plan->bytes = 2.0 * (static_cast<double>(g.N) * p.srcH * p.srcW * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC + static_cast<double>(g.OC) * g.IC * 9);
    char buf[256];
    snprintf(buf, sizeof(buf), "conv2d_mfma_wide_f16_32x32x16 k=3x3 s=1 ic=%d oc=%d tile=%dx32px x %doc (4x%d MFMA tiles per wave) chunk=%d lds=%zuB", g.IC, g.OC, TH, BN, NT,
             16 * C8, lds);
    plan->desc = buf;
    if (g.preMode) plan->desc += " +pad(" + std::string(g.preMode == SNNHIP_PAD_REFLECT ? "reflect" : g.preMode == SNNHIP_PAD_REPLICATE ? "replicate" : "constant") + ")";
    if (g.preMode && g.preShift) plan->desc += " +upsample(x2)";
    if (g.normShift) plan->desc = "instancenorm(act=" + std::to_string(g.normAct) + ", in LDS behind the DMA) -> " + plan->desc;
    if (plan->fusedAdd) {
        plan->desc += " +add";
        plan->bytes += 2.0 * static_cast<double>(g.N) * g.OH * g.OW * g.OC;
    }
    if (plan->p.kTok) plan->desc += " +ktoken";
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
