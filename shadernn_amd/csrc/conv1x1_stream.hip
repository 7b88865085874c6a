// conv1x1_stream.hip -- pointwise (1x1, stride 1 or 2) convolution as a streaming GEMM on v_mfma_f32_32x32x2_f32 (fp32 tensors) or
// v_mfma_f32_32x32x16_f16 (half tensors, see conv1x1_stream_f16_kernel), for the layers whose time is
// their activation / output stream rather than their arithmetic: MobileNetV2's expand (16->96 ... 96->576) and project (96->24 ... 192->32)
// convolutions at 112x112 .. 28x28 (BASELINE configs[3]).  Replaces shadertemplate_vk_conv2d_1x1.comp:68-210 of the reference for those
// shapes; bias -> BN -> activation epilogue and the fused residual Add are those of conv2d_mfma_kernel (same helpers, same rounding points).
//
// Why a second kernel: for a 1x1 stride-1 layer the NHWC tensor IS the row-major GEMM operand (M = N*H*W pixels, K = IC contiguous), so the
// halo-tile machinery of conv2d_mfma_kernel (LDS staging, a barrier per chunk, a prologue / epilogue per 128-pixel block) is pure overhead:
// on 16->96 @112x112 b32 it reached 2.9 TB/s of unfused traffic, 40 % of what `tools/ubench_hbm.hip` measures for a copy.  Here
//   * a wave owns one 32-pixel row tile; its A operands come straight from global memory in the MFMA's own layout (lane (row = l%32,
//     h = l/32) loads the 16 bytes x[row][8c + 4h .. +3] of chunk c: the K order inside an 8-channel chunk is permuted so that four
//     consecutive MFMAs consume the four components, exactly the packing conv2d_mfma uses), 4-8 chunks in flight per lane -- no LDS, no
//     barrier on the activation path, and thousands of short independent waves for the memory system to overlap;
//   * the block's weight slice [IC][32*NT oc] is staged ONCE per block in LDS in that packed order (one ds_read_b128 per chunk and MFMA column);
//   * D: lane holds column oc = l%32 and rows 8g + 4h + k, so a store instruction writes 32 consecutive channels (128 B) of two pixels and
//     the NT column tiles of a pixel are written back to back (a full 384-byte row for 96 channels).
#include "epilogue.h"
#include "snnhip_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace snnhip {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct StreamParams {
    int M;       // pixels = N*H*W
    int IC, OC;
    int nChunks; // IC / 8
    int phaseChunks; // chunks of the weight slice resident in LDS at a time (= nChunks unless the slice is staged in several K phases; multiple of 8)
    int nTiles;  // ceil(M / 32)
    int useBN;
    int stride, W, HW, OW, OHW; // stride > 1 (ResNet's 1x1 s2 downsample convolutions): output row -> input pixel (n, s*oy, s*ox); HW = H*W, OHW = OH*OW
    const float* res; // fused residual Add (chain rule E), set per launch
    ActCfg ac2;
    int segTiles; // conv1x1_march_kernel (round 4): 32-pixel tiles per block segment
};

// kDepth = activation chunks (8 channels = 16 bytes per lane each) in flight per lane: 4 next to 48 accumulators, 8 for the one-column
// project layers (deep K, everything they move is the activation read)
// kStreamWaves = 32-pixel tiles (waves) per block = pixels per staged weight slice / 32.  4 for the layers that narrow (project: OC < 3 IC), 8 for the
// ones that widen (MobileNetV2's expand layers and head): their slices are the big ones (96 output channels x IC per block, re-staged for every
// 128 pixels: 86 MB of L2 -> LDS traffic next to 115 MB of output on 96 -> 576 at 14x14), and a block of eight waves stages a slice for 256
// pixels with half the work per thread.  Per layer at batch 256, 4 / 8 waves, us: 160 -> 960 60 / 50, 320 -> 1280 120 / 113, 96 -> 576 88 / 86,
// 64 -> 384 51 / 49; the project layers lose (384 -> 96 55 / 62, 576 -> 160 36 / 38) and keep four.
template <int NT, bool SIMPLE, int kStreamWaves>
__global__ __launch_bounds__(64 * kStreamWaves) void conv1x1_stream_kernel(StreamParams p, ActCfg ac, const float* __restrict__ x, const float4* __restrict__ wp,
                                                          const float4* __restrict__ epi, float* __restrict__ y) {
    extern __shared__ float4 s_w[]; // [nChunks][2][BN]
    constexpr int BN = 32 * NT;
    constexpr int kStreamThreads = 64 * kStreamWaves;
    constexpr int kDepth = NT == 1 ? 8 : 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
#ifdef SNNHIP_STREAM_TRACE // experiment builds (tools/exp_one.sh)
    const bool strace = blockIdx.x == 40 && blockIdx.y == 1 && lane == 0 && (wave == 0 || wave == 3);
    unsigned long long sstamp[5] = {};
    if (strace) sstamp[0] = __builtin_readcyclecounter();
#endif
    float4 e[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) e[u] = epi[n0 + u * 32 + l32]; // table padded to the block grid's channel count

    // one 32-pixel tile per wave.  (A persistent walk over several tiles per wave, with the next tile's first chunks requested before the
    // epilogue, measured 10-100 % SLOWER the more tiles a wave owned: these layers live on memory-level parallelism, and a wave that is busy
    // with its 48-store epilogue is not issuing loads -- many short waves keep more requests in flight than few long ones.  Two tiles per wave
    // side by side, sharing the LDS weight reads, also lost: 144->24 @56x56 b32 22.0 -> 23.6 us, 192->32 @28x28 9.2 -> 16.6 us.  So did
    // requesting the first activation chunks BEFORE the weight staging above: vmcnt retires in order, so the L2-resident weights then wait for
    // the HBM loads in front of them and the barrier moves out (16->96 @112x112 36.4 -> 42.2 us).)
    const int tile = blockIdx.x * kStreamWaves + wave;
    const bool active = tile < p.nTiles; // (wave-uniform; an idle wave of the last block still takes part in the barriers of the weight phases)
    // chunk c of the tile's row l32 for this lane's K half; rows past M and chunks past IC read as zero (their products vanish / are not stored)
    const int arow = tile * 32 + l32;
    int irow = arow < p.M ? arow : 0;
    if (p.stride > 1) { // the lane addresses its own row anyway, so a strided layer is one index decode per lane, not a different access pattern
        const unsigned n = static_cast<unsigned>(irow) / static_cast<unsigned>(p.OHW), rem = static_cast<unsigned>(irow) - n * p.OHW;
        const unsigned oy = rem / static_cast<unsigned>(p.OW), ox = rem - oy * p.OW;
        irow = static_cast<int>(n * p.HW + (oy * p.W + ox) * p.stride);
    }
    const float* xrow = x + static_cast<size_t>(irow) * p.IC + h * 4;
    auto loadA = [&](int c) -> float4 {
        if (arow < p.M && c < p.nChunks) return *reinterpret_cast<const float4*>(xrow + c * 8);
        return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    };
    float4 nxt[kDepth];
#pragma unroll
    for (int d = 0; d < kDepth; ++d) nxt[d] = loadA(d);
    f32x16 acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[u][i] = 0.0f;
    // The weight slice [IC][BN] sits in LDS in K PHASES of p.phaseChunks chunks (one phase for the slices that fit: every MobileNetV2 layer up to 576
    // input channels; the 960-channel layers of the 7x7 stage take two -- they used to fall to the 128-pixel tile kernel at 0.22-0.34 of the roofline).
    for (int pb = 0; pb < p.nChunks; pb += p.phaseChunks) {
        const int pe = min(p.nChunks, pb + p.phaseChunks);
        if (pb) __syncthreads(); // every wave is done with the previous phase's weights
        {   // this phase's weights -> LDS, eight loads in flight per thread (one at a time -- load, wait, store -- the staging of a 320 x 64 slice took
            // 21 000 cycles of a 90 000-cycle block: an L2 round trip per 16 bytes)
            const int cnt = (pe - pb) * 2 * BN;
            const float4* src = wp + (static_cast<size_t>(blockIdx.y) * p.nChunks + pb) * 2 * BN;
            for (int i0 = tid; i0 < cnt; i0 += kStreamThreads * 8) {
                float4 t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = src[min(i0 + kStreamThreads * j, cnt - 1)]; // (unconditional: a partly written register array goes to scratch)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (i0 + kStreamThreads * j < cnt) s_w[i0 + kStreamThreads * j] = t[j];
            }
        }
        __syncthreads();
#ifdef SNNHIP_STREAM_TRACE
        if (strace && pb == 0) sstamp[1] = __builtin_readcyclecounter();
#endif
        if (!active) continue;
        for (int c0 = pb; c0 < pe; c0 += kDepth) {
            float4 cur[kDepth];
#pragma unroll
            for (int d = 0; d < kDepth; ++d) cur[d] = nxt[d];
            if (c0 + kDepth < p.nChunks) {
#pragma unroll
                for (int d = 0; d < kDepth; ++d) nxt[d] = loadA(c0 + kDepth + d);
            }
#pragma unroll
            for (int d = 0; d < kDepth; ++d) {
                if (c0 + d < pe) { // wave-uniform
                    float4 b[NT];
#pragma unroll
                    for (int u = 0; u < NT; ++u) b[u] = s_w[((c0 + d - pb) * 2 + h) * BN + u * 32 + l32];
#pragma unroll
                    for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[d].x, b[u].x, acc[u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[d].y, b[u].y, acc[u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[d].z, b[u].z, acc[u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[d].w, b[u].w, acc[u], 0, 0, 0);
                }
            }
        }
    }
    if (!active) return;
#ifdef SNNHIP_STREAM_TRACE
    if (strace) sstamp[2] = __builtin_readcyclecounter();
#endif
    // epilogue: bias -> BN -> activation [-> + residual -> activation of the Add layer]
    const int row0 = tile * 32 + 4 * h;
    const bool addSimple = act_is_simple_dev(p.ac2.act);
    if ((p.OC & 3) == 0) {
        // A lane holds ONE channel (column l32) of 16 pixels; written as it stands that is 16 NT scalar stores per lane (48 for a 96-channel tile: the
        // phase trace of MobileNetV2's expand layers had the epilogue as long as the K loop).  The four values of a channel run (pixels R .. R + 3) are
        // transposed across the four lanes of a quad (channels 4m .. 4m + 3) with two DPP exchange steps: lane q then holds pixel R + q, channels
        // 4m .. 4m + 3 -- ONE 16-byte store, the quads of a row side by side (whole 128-byte lines), a quarter of the store instructions; the residual
        // of a fused Add arrives the same way, as 16-byte loads that are all requested before the first store.
        const int q = l32 & 3, m4 = l32 & ~3;
        const bool qb0 = (q & 1) != 0, qb1 = (q & 2) != 0;
        float4 rv[NT][4];
        if (p.res) {
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = row0 + 8 * g + q, c0 = n0 + u * 32 + m4;
                    rv[u][g] = (row < p.M && c0 < p.OC) ? *reinterpret_cast<const float4*>(p.res + static_cast<size_t>(row) * p.OC + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = epi_affine(acc[u][4 * g + k], e[u], p.useBN);
                    v[k] = SIMPLE ? apply_act<true>(ac, t, 0.0f) : epi_act(ac.act, ac.leaky, t, 0.0f);
                }
                // 4 x 4 transpose (register index k <-> lane index q of the quad): swap bit 0 with the lane one over, then bit 1 with the lane two over
                float x[4], t1[4], o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[k]), 0xB1, 0xF, 0xF, true)); // quad_perm [1, 0, 3, 2]
                t1[0] = qb0 ? x[1] : v[0];
                t1[1] = qb0 ? v[1] : x[0];
                t1[2] = qb0 ? x[3] : v[2];
                t1[3] = qb0 ? v[3] : x[2];
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t1[k]), 0x4E, 0xF, 0xF, true)); // quad_perm [2, 3, 0, 1]
                o[0] = qb1 ? x[2] : t1[0];
                o[1] = qb1 ? x[3] : t1[1];
                o[2] = qb1 ? t1[2] : x[0];
                o[3] = qb1 ? t1[3] : x[1];
                const int row = row0 + 8 * g + q, c0 = n0 + u * 32 + m4;
                if (row < p.M && c0 < p.OC) {
                    if (p.res) {
                        o[0] = add_act(p.ac2, addSimple, o[0] + rv[u][g].x);
                        o[1] = add_act(p.ac2, addSimple, o[1] + rv[u][g].y);
                        o[2] = add_act(p.ac2, addSimple, o[2] + rv[u][g].z);
                        o[3] = add_act(p.ac2, addSimple, o[3] + rv[u][g].w);
                    }
                    *reinterpret_cast<float4*>(y + static_cast<size_t>(row) * p.OC + c0) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = row0 + 8 * g + k;
                if (row < p.M) {
                    const size_t o = static_cast<size_t>(row) * p.OC + n0 + l32;
#pragma unroll
                    for (int u = 0; u < NT; ++u) {
                        if (n0 + u * 32 + l32 < p.OC) {
                            float v = epi_affine(acc[u][4 * g + k], e[u], p.useBN);
                            v = SIMPLE ? apply_act<true>(ac, v, 0.0f) : epi_act(ac.act, ac.leaky, v, 0.0f);
                            if (p.res) v = add_act(p.ac2, addSimple, v + p.res[o + u * 32]);
                            y[o + u * 32] = v;
                        }
                    }
                }
            }
    }
#ifdef SNNHIP_STREAM_TRACE
    if (strace)
        printf("streamtrace ic %d oc %d NT %d M %d res %d wave %d: weights+barrier %llu kloop %llu epilogue %llu\n", p.IC, p.OC, NT, p.M, p.res != nullptr, wave, sstamp[1] - sstamp[0],
               sstamp[2] - sstamp[1], __builtin_readcyclecounter() - sstamp[2]);
#endif
}

// s_waitcnt vmcnt(n) for a compile-time n (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
#ifdef SNNHIP_M1_TRACE // experiment builds (tools/exp_one.sh): one block prints the s_memtime stamps of its phases
#define M1_MARK(i) do { if (mtrace || ((i) == 4 && mtotal_pre)) mstamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define M1_MARK(i) do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt_n() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// conv1x1_march_kernel (round 4) -- the WIDENING pointwise layers (MobileNetV2's expand convolutions 64 -> 384, 96 -> 576, 160 -> 960 at batch 256: 0.44 of
// their roofline on the kernel above, whose waves each load their own activations, wait for them, multiply, and then spend as long again in a 48-store
// epilogue during which they request nothing).  Same arithmetic, same packed weights, other roles:
//   * a block is PERSISTENT on one 96-channel output block and a run of 32-pixel tiles: its weight slice [IC][96] goes to LDS ONCE (the kernel above
//     re-stages it for every 128 / 256 pixels: 86 MB of L2 -> LDS traffic next to 115 MB of output on 96 -> 576);
//   * ONE LOADER wave copies the activation tiles of the NEXT step into an LDS ring with global_load_lds_dwordx4 while the compute waves work on this
//     step's (rows IC / 4 + 1 16-byte slots apart -- odd -- so the 32 rows of an operand read fall into different bank groups); it never stores, so its
//     vmcnt counts loads only;
//   * 3 TS COMPUTE waves: wave (tile ts, column tile u) multiplies tile ts by the 32 output channels u of the slice -- 16 accumulators, both operands
//     one ds_read_b128 per 8-channel chunk -- and stores its 32 x 32 result, one 128-byte pixel run per half wave and accumulator; three waves per SIMD
//     interleave K loops and epilogues;
//   * one s_barrier per step (TS tiles).
template <int TS, int PI, int EPI /* 0: any activation; 1: none / relu / relu6 / leakyRelu; 2: those without the leaky slope */>
__global__ __launch_bounds__(64 * (3 * TS + 1), (3 * TS + 4) / 4) void conv1x1_march_kernel(StreamParams p, ActCfg ac, const float* __restrict__ x, const float4* __restrict__ wp,
                                                                                             const float4* __restrict__ epi, float* __restrict__ y) {
    extern __shared__ float4 s_w[]; // [nChunks][2][96] weights, then the ring: 2 halves x TS tiles x (PI * 64) slots
    constexpr int BN = 96, NCW = 3 * TS, kTileSlots = PI * 64;
    static_assert(TS * PI <= 63, "one step of tiles in flight: vmcnt is a 6-bit counter");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l32 = lane & 31, h = lane >> 5; // (wave: a scalar for the compiler)
    const int n0 = blockIdx.y * BN;
    const int SP = p.IC / 4 + 1; // 16-byte slots per pixel row of a tile (the last one padding: IC / 4 is even, the row pitch odd)
    float4* const ringp = s_w + p.nChunks * 2 * BN;
    const int t0 = blockIdx.x * p.segTiles, t1 = min(p.nTiles, t0 + p.segTiles);
    const int steps = (t1 - t0 + TS - 1) / TS;
    if (steps <= 0) return; // (block-uniform)
#ifdef SNNHIP_M1_TRACE
    const bool mtrace = blockIdx.x == 10 && blockIdx.y == 1 && lane == 0 && (wave == 0 || wave == 5 || wave == NCW);
    unsigned long long mstamp[5] = {};
    const bool mtotal_pre = true;
    M1_MARK(4);
    const unsigned long long mwall0 = wall_clock64();
    const bool mtotal = (blockIdx.x % 20 == 3) && lane == 0 && wave == 0; // un-perturbed blocks: whole-block cycles and the 100 MHz wall clock beside them
#endif

    if (wave == NCW) { // ---- loader
        // per lane and copy: the byte offset of its 16 bytes inside a tile's 32 x IC block.  The lanes of the padding slot of a row (and those past the 32 rows in
        // the last copy) re-read the tile's first 16 bytes: no compute wave reads what they write, and the copy then needs neither an exec mask nor a
        // select -- the loader's steady state is SALU + the copies themselves.
        unsigned voff[PI];
#pragma unroll
        for (int k = 0; k < PI; ++k) {
            const int e = 64 * k + lane, row = e / SP, sl = e - row * SP;
            voff[k] = (row < 32 && sl < SP - 1) ? static_cast<unsigned>(row * p.IC + sl * 4) * 4u : 0u;
        }
        const unsigned ringLds = lds_byte_addr(ringp);
        auto issue_step = [&](int st) {
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) {
                const int tile = t0 + st * TS + ts; // (wave-uniform)
                if (tile >= t1) break;
                const float* xt = x + static_cast<size_t>(tile) * 32 * p.IC;
                const unsigned dst = ringLds + static_cast<unsigned>(((st & 1) * TS + ts) * kTileSlots) * 16u;
                if (tile * 32 + 32 <= p.M) {
#pragma unroll
                    for (int k = 0; k < PI; ++k) lds_dma16_sbase(xt, voff[k], dst + 1024u * k);
                } else { // the tensor's last, partial tile: rows past the end re-read the tile's first bytes (their results are not stored)
                    const int lastRow = p.M - 1 - tile * 32;
#pragma unroll
                    for (int k = 0; k < PI; ++k) lds_dma16_sbase(xt, (64 * k + lane) / SP <= lastRow ? voff[k] : 0u, dst + 1024u * k);
                }
            }
        };
        issue_step(0);
        for (int st = 0; st < steps; ++st) {
            M1_MARK(0);
            wait_vmcnt_n<0>();
            M1_MARK(1);
            __syncthreads(); // step st is published; every compute wave has left step st - 1 (whose half of the ring step st + 1 overwrites)
            M1_MARK(2);
            if (st + 1 < steps) issue_step(st + 1);
            M1_MARK(3);
#ifdef SNNHIP_M1_TRACE
            if (mtrace && st >= 2 && st < 5)
                printf("m1 loader st %d: wait-loads %llu barrier %llu issue %llu\n", st, mstamp[1] - mstamp[0], mstamp[2] - mstamp[1], mstamp[3] - mstamp[2]);
#endif
        }
        wait_vmcnt_n<0>();
        return;
    }

    // ---- compute waves: the weight slice once, then one tile x 32 channels per step
    {
        const int cnt = p.nChunks * 2 * BN;
        const float4* src = wp + static_cast<size_t>(blockIdx.y) * cnt;
        for (int i0 = tid; i0 < cnt; i0 += 64 * NCW * 4) {
            float4 t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = src[min(i0 + 64 * NCW * j, cnt - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + 64 * NCW * j < cnt) s_w[i0 + 64 * NCW * j] = t[j];
        }
    }
    const int ts = wave / 3, u = wave - 3 * ts;
    const float4 e = epi[n0 + u * 32 + l32];
    const float4* const bop = s_w + h * BN + u * 32 + l32; // + chunk * 2 * BN
    for (int st = 0; st < steps; ++st) {
        M1_MARK(0);
        __syncthreads();
        M1_MARK(1);
        const int tile = t0 + st * TS + ts;
        if (tile >= t1) continue; // (wave-uniform; the barrier of the next step is at the loop top)
        const float4* const aop = ringp + ((st & 1) * TS + ts) * kTileSlots + l32 * SP + h; // + 2 * chunk
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        float4 a = aop[0], b = bop[0];
        for (int c = 0; c < p.nChunks; ++c) { // the next chunk's operands are on their way while this chunk's four MFMAs issue
            const int cn = min(c + 1, p.nChunks - 1);
            const float4 an = aop[2 * cn], bn = bop[cn * 2 * BN];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            a = an;
            b = bn;
        }
        M1_MARK(2);
        // epilogue: bias -> BN -> activation, ONE 4-byte store per accumulator: a half wave's lanes are 32 channels of one pixel (128 contiguous bytes, a whole
        // line), the row base is scalar and the lane offset a constant, so a store costs no VALU instruction.  (The quad-transposed 16-byte stores of
        // conv1x1_stream_kernel cost 4 VALU per value here: 64 of a wave's 176 epilogue instructions, on the pipe the MFMAs need.  36.0 vs 37.9 us on 64 -> 384.)
        {
            const unsigned vo = static_cast<unsigned>(4 * h * p.OC + u * 32 + l32) * 4u;
            float* const yt = y + static_cast<size_t>(tile) * 32 * p.OC + n0;
            auto value = [&](int i, auto bnTag) -> float {
                float t = acc[i] + e.x;
                if (decltype(bnTag)::value) t = (e.y * (t - e.z)) + e.w; // (epi_affine)
                if (EPI == 2) return __builtin_amdgcn_fmed3f(t, ac.lo, ac.hi); // alpha == 1: apply_act<true>'s fmaxf(t, t * 1) is t
                return EPI == 1 ? apply_act<true>(ac, t, 0.0f) : epi_act(ac.act, ac.leaky, t, 0.0f);
            };
            auto emit = [&](auto bnTag) {
                if (tile * 32 + 32 <= p.M) { // (wave-uniform)
#pragma unroll
                    for (int i = 0; i < 16; ++i) store_dword_sbase(yt + static_cast<size_t>(8 * (i >> 2) + (i & 3)) * p.OC, vo, value(i, bnTag));
                } else { // the tensor's last, partial tile
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int r = 8 * (i >> 2) + (i & 3);
                        if (tile * 32 + r + 4 * h < p.M) yt[static_cast<size_t>(r) * p.OC + (vo >> 2)] = value(i, bnTag);
                    }
                }
            };
            if (p.useBN) emit(std::true_type{});
            else emit(std::false_type{});
        }
        M1_MARK(3);
#ifdef SNNHIP_M1_TRACE
        if (mtotal && st == steps - 1)
            printf("m1 block (%d,%d) total: %llu cycles, %llu wall ticks (100 MHz), steps %d\n", blockIdx.x, blockIdx.y, __builtin_readcyclecounter() - mstamp[4],
                   wall_clock64() - mwall0, steps);
        if (mtrace && st >= 2 && st < 5)
            printf("m1 wave %d st %d: barrier %llu k-loop %llu epilogue %llu (steps %d, block start->now %llu)\n", wave, st, mstamp[1] - mstamp[0], mstamp[2] - mstamp[1], mstamp[3] - mstamp[2], steps,
                   mstamp[3] - mstamp[4]);
#endif
    }
}

// fp16 tensors (half storage, fp32 accumulation on v_mfma_f32_32x32x16_f16): a chunk is 16 channels, lane (row, h) loads the 16 bytes
// x[row][16c + 8h .. +7] and ONE MFMA consumes them.  A lane's results are single halfs of 16 different pixels; written directly they would
// leave as 2-byte stores in 64-byte runs, so every wave transposes its 32 x BN tile through its own LDS region (no block barrier: the wave
// only reads what it wrote itself) and stores 16-byte vectors, a pixel's BN channels contiguous.  The fused Add is applied there, on the
// rounded convolution result, with conv2d_mfma's rounding points.  Needs OC % 8 == 0.
template <int NT, bool SIMPLE>
__global__ __launch_bounds__(256) void conv1x1_stream_f16_kernel(StreamParams p, ActCfg ac, const float* __restrict__ xv, const float4* __restrict__ wp,
                                                              const float4* __restrict__ epi, float* __restrict__ yv) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    extern __shared__ float4 s_w[]; // [nChunks][2][BN] x 8 halfs, then 4 output tiles of 32 x (BN + 8) halfs
    constexpr int BN = 32 * NT, EP = BN + 8;
    constexpr int kDepth = NT == 1 ? 8 : 4;
    const _Float16* __restrict__ x = reinterpret_cast<const _Float16*>(xv);
    _Float16* __restrict__ y = reinterpret_cast<_Float16*>(yv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
    {   // the block's weight slice -> LDS, eight loads in flight per thread (one at a time -- load, wait, store -- the staging of a 320 x 64 slice took
        // 21 000 cycles of a 90 000-cycle block: an L2 round trip per 16 bytes)
        const int cnt = p.nChunks * 2 * BN;
        const float4* src = wp + static_cast<size_t>(blockIdx.y) * cnt;
        for (int i0 = tid; i0 < cnt; i0 += 256 * 8) {
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = src[min(i0 + 256 * j, cnt - 1)]; // (unconditional: a partly written register array goes to scratch)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (i0 + 256 * j < cnt) s_w[i0 + 256 * j] = t[j];
        }
    }
    float4 e[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) e[u] = epi[n0 + u * 32 + l32];
    __syncthreads();
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= p.nTiles) return;
    const int arow = tile * 32 + l32;
    int irow = arow < p.M ? arow : 0;
    if (p.stride > 1) {
        const unsigned n = static_cast<unsigned>(irow) / static_cast<unsigned>(p.OHW), rem = static_cast<unsigned>(irow) - n * p.OHW;
        const unsigned oy = rem / static_cast<unsigned>(p.OW), ox = rem - oy * p.OW;
        irow = static_cast<int>(n * p.HW + (oy * p.W + ox) * p.stride);
    }
    const _Float16* xrow = x + static_cast<size_t>(irow) * p.IC + h * 8;
    auto loadA = [&](int c) -> float4 { // IC % 16 == 8: the upper half of the last chunk lies past the row and reads as zero
        if (arow < p.M && c * 16 + h * 8 < p.IC) return *reinterpret_cast<const float4*>(xrow + c * 16);
        return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    };
    float4 nxt[kDepth];
#pragma unroll
    for (int d = 0; d < kDepth; ++d) nxt[d] = loadA(d);
    f32x16 acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[u][i] = 0.0f;
    for (int c0 = 0; c0 < p.nChunks; c0 += kDepth) {
        float4 cur[kDepth];
#pragma unroll
        for (int d = 0; d < kDepth; ++d) cur[d] = nxt[d];
        if (c0 + kDepth < p.nChunks) {
#pragma unroll
            for (int d = 0; d < kDepth; ++d) nxt[d] = loadA(c0 + kDepth + d);
        }
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
            if (c0 + d < p.nChunks) { // wave-uniform
                float4 b[NT];
#pragma unroll
                for (int u = 0; u < NT; ++u) b[u] = s_w[((c0 + d) * 2 + h) * BN + u * 32 + l32];
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&cur[d]), *reinterpret_cast<const h8*>(&b[u]), acc[u], 0, 0, 0);
            }
        }
    }
    // epilogue: bias -> BN -> activation, rounded to half into the wave's own LDS tile [row][BN]
    _Float16* const ot = reinterpret_cast<_Float16*>(s_w + p.nChunks * 2 * BN) + wave * 32 * EP;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 8 * g + 4 * h + k;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                float v = epi_affine(acc[u][4 * g + k], e[u], p.useBN);
                v = SIMPLE ? apply_act<true>(ac, v, 0.0f) : epi_act(ac.act, ac.leaky, v, 0.0f);
                ot[r * EP + u * 32 + l32] = static_cast<_Float16>(v);
            }
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int VPR = BN / 8; // 16-byte vectors per row
    const bool addSimple = act_is_simple_dev(p.ac2.act);
    const _Float16* res = reinterpret_cast<const _Float16*>(p.res);
#pragma unroll
    for (int j = 0; j < 32 * VPR / 64; ++j) {
        const int vi = lane + 64 * j;
        const int r = vi / VPR, c8 = vi - r * VPR;
        const int row = tile * 32 + r, oc = n0 + c8 * 8;
        if (row < p.M && oc < p.OC) {
            const size_t o = static_cast<size_t>(row) * p.OC + oc;
            float4 pack = *reinterpret_cast<const float4*>(ot + r * EP + c8 * 8);
            if (res) {
                const float4 rpack = *reinterpret_cast<const float4*>(res + o);
                const _Float16* ch = reinterpret_cast<const _Float16*>(&pack);
                const _Float16* rh = reinterpret_cast<const _Float16*>(&rpack);
                _Float16 oh[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) oh[q] = static_cast<_Float16>(add_act(p.ac2, addSimple, static_cast<float>(ch[q]) + static_cast<float>(rh[q])));
                pack = *reinterpret_cast<const float4*>(oh);
            }
            *reinterpret_cast<float4*>(y + o) = pack;
        }
    }
}

struct Conv1x1StreamPlan : ConvPlanBase {
    StreamParams p;
    ActCfg ac;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    size_t ldsBytes = 0;
    dim3 grid;
    void (*kernel)(StreamParams, ActCfg, const float*, const float4*, const float4*, float*) = nullptr;
    bool fusedAdd = false;
    int waves = 4; // 32-pixel tiles per block

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == (fusedAdd ? 2 : 1), "conv2d: expects %d input(s), got %d", fusedAdd ? 2 : 1, nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == g.N && x->h == g.H && x->w == g.W && x->c == g.IC, "conv2d: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h, x->w, x->c,
                       g.N, g.H, g.W, g.IC);
        SNNHIP_REQUIRE(out->n == g.N && out->h == g.OH && out->w == g.OW && out->c == g.OC, "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n,
                       out->h, out->w, out->c, g.N, g.OH, g.OW, g.OC);
        StreamParams q = p;
        q.res = nullptr;
        if (fusedAdd) {
            const snnhip_tensor* r = in[1];
            SNNHIP_REQUIRE(r->n == g.N && r->h == g.OH && r->w == g.OW && r->c == g.OC && r->dtype == dtype,
                           "conv2d+add: residual %dx%dx%dx%d (dtype %d) does not match the output %dx%dx%dx%d", r->n, r->h, r->w, r->c, r->dtype, g.N, g.OH, g.OW,
                           g.OC);
            q.res = r->data;
        }
        SNNHIP_LAUNCH(kernel, grid, dim3(64 * waves), ldsBytes, ctx->stream, q, ac, static_cast<const float*>(x->data), reinterpret_cast<const float4*>(d_w),
                           reinterpret_cast<const float4*>(d_epi), out->data);
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

template <int NT>
decltype(Conv1x1StreamPlan::kernel) pick(bool simple, int waves) {
    if (waves == 8) return simple ? conv1x1_stream_kernel<NT, true, 8> : conv1x1_stream_kernel<NT, false, 8>;
    return simple ? conv1x1_stream_kernel<NT, true, 4> : conv1x1_stream_kernel<NT, false, 4>;
}
// conv1x1_march_kernel instantiation for `pieces` 1 KB DMA pieces per 32-pixel tile: four tiles per step where that is at most 63 pieces and the ring fits, else two
decltype(Conv1x1StreamPlan::kernel) pick_march(int pieces, int epi, int* TS, int* PI) {
#define SNNHIP_MARCH(T, P) (*TS = T, *PI = P, epi == 2 ? conv1x1_march_kernel<T, P, 2> : epi == 1 ? conv1x1_march_kernel<T, P, 1> : conv1x1_march_kernel<T, P, 0>)
    if (pieces <= 5) return SNNHIP_MARCH(4, 5);
    if (pieces <= 9) return SNNHIP_MARCH(4, 9);
    if (pieces <= 13) return SNNHIP_MARCH(4, 13);
    if (pieces <= 17) return SNNHIP_MARCH(2, 17);
    if (pieces <= 21) return SNNHIP_MARCH(2, 21);
#undef SNNHIP_MARCH
    return nullptr;
}
template <int NT>
decltype(Conv1x1StreamPlan::kernel) pick16(bool simple) {
    return simple ? conv1x1_stream_f16_kernel<NT, true> : conv1x1_stream_f16_kernel<NT, false>;
}

} // namespace

// SNNHIP_E_UNSUPPORTED when the layer is not a large fp32 pointwise stream (the caller then takes the general MFMA kernel).
// SNNHIP_CONV_1X1=0 switches the kernel off (A/B runs, tests of the general kernel on the same shapes), =2 also takes few-tile deep-K layers.
int make_conv1x1_stream_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    const char* sw = snnhip::option("SNNHIP_CONV_1X1"); // 0: off; 2: also take few-tile deep-K layers (parity tests at oracle-sized shapes)
    if (sw && atoi(sw) == 0) return SNNHIP_E_UNSUPPORTED;
    if (g.normShift) return SNNHIP_E_UNSUPPORTED; // graph rule I: not in this kernel
    const bool f16 = g.dtype == SNNHIP_F16;
    if (f16 && g.OC % 8 != 0) return SNNHIP_E_UNSUPPORTED; // the fp16 variant stores 8-channel vectors
    if ((g.dtype != SNNHIP_F32 && !f16) || g.kh != 1 || g.kw != 1 || g.sh != g.sw || g.sh < 1 || g.sh > 2 || g.preMode != 0 || g.padx != 0 || g.pady != 0) return SNNHIP_E_UNSUPPORTED;
    if ((g.OH - 1) * g.sh >= g.H || (g.OW - 1) * g.sw >= g.W || (g.sh == 1 && (g.OH != g.H || g.OW != g.W))) return SNNHIP_E_UNSUPPORTED;
    if (g.act == SNNHIP_ACT_SILU_QUIRK || g.IC < 8 || g.IC % 8 != 0 || g.OC < 16) return SNNHIP_E_UNSUPPORTED;
    const double M = static_cast<double>(g.N) * g.OH * g.OW;
    if (static_cast<double>(g.N) * g.H * g.W * g.IC >= 2147483647.0 || M * g.OC >= 2147483647.0) return SNNHIP_E_UNSUPPORTED;
    const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
    const int nTiles = static_cast<int>((static_cast<long long>(M) + 31) / 32);
    // measured against the general kernel (tools/bench_layers.py --shape ..., SNNHIP_CONV_1X1=0 / 2): faster or equal on every MobileNetV2 /
    // YOLO pointwise layer down to 7x7 maps (160->960 @7x7 b32: 25.0 -> 13.8 us) whose weight slice fits 80 KB of LDS; beyond that (576 ->
    // 96+, 960 -> 160 ...) two resident blocks per CU are too few and the general kernel's split-K wins, as it does when a deep reduction
    // meets a handful of tiles (checked after the column width is known)
    // columns per block: the widest of 96 / 64 / 32 that pads the channel count by at most a third and whose weight slice fits the LDS budget
    size_t ldsCap = 80 * 1024;
    if (const char* e = snnhip::option("SNNHIP_CONV_1X1_LDS_KB")) ldsCap = static_cast<size_t>(atoi(e)) * 1024; // experiments
    int NT = 0, KP = 1;
    for (int c = 3; c >= 1 && !NT; --c) {
        const int ocp = (g.OC + 32 * c - 1) / (32 * c) * (32 * c);
        const size_t wBytes = f16 ? static_cast<size_t>((g.IC + 15) / 16) * 2 * 32 * c * 16 + static_cast<size_t>(4) * 32 * (32 * c + 8) * 2
                                  : static_cast<size_t>(g.IC) * 32 * c * 4;
        if ((c == 1 || (ocp - g.OC) * 3 <= g.OC) && wBytes <= ldsCap) NT = c;
    }
    // fp32 slices that do not fit (MobileNetV2's 960 -> 160 / 320 at 7x7): staged in two K phases of <= 64 KB (two blocks per CU), from a few thousand
    // pixel rows on (the few-tile layers of a small batch stay with the general kernel's split-K); SNNHIP_CONV_1X1_PHASES=0 switches it off
    if (!NT && !f16 && M >= 8192.0 && !(snnhip::option("SNNHIP_CONV_1X1_PHASES") && atoi(snnhip::option("SNNHIP_CONV_1X1_PHASES")) == 0)) {
        for (int c = 2; c >= 1 && !NT; --c) {
            const int ocp = (g.OC + 32 * c - 1) / (32 * c) * (32 * c);
            const size_t wBytes = static_cast<size_t>(g.IC) * 32 * c * 4;
            if ((c == 1 || (ocp - g.OC) * 3 <= g.OC) && wBytes <= 2 * 64 * 1024) {
                NT = c;
                KP = 2;
            }
        }
    }
    if (!NT) return SNNHIP_E_UNSUPPORTED;
    // few-tile layers (ResNet-18's 1x1 stride-2 downsample convolutions at batch 32: 196 / 49 row tiles): a wave's K loop is a serial chain of IC / 8 x 4 NT
    // MFMAs, so the widest column leaves a few hundred long waves on 1024 SIMDs.  Narrower columns until the grid has three waves per CU (the activations are
    // re-read from L2, which these layers do not notice).  tools/r6_ds.sh, batch 32, NT 3 / 2 / 1: 128->256 @28x28 13.0 / 11.8 / 11.6 us, 256->512 @14x14 - / 18.2 / 10.7 us
    const char* marchPin = snnhip::option("SNNHIP_CONV_1X1_MARCH"); // (=1 forces the 96-channel persistent form on small layers: tests; it needs the full width)
    if (!f16 && !(marchPin && atoi(marchPin) == 1))
        while (NT > 1 && static_cast<long long>(nTiles) * ((g.OC + 32 * NT - 1) / (32 * NT)) < 3LL * cus) --NT;
    if (const char* pin = snnhip::option("SNNHIP_CONV_1X1_NT")) // experiments: pin the column width (1-3 x 32 channels)
        if (atoi(pin) >= 1 && atoi(pin) <= 3) NT = atoi(pin);
    // a slice that fits but leaves one or two blocks per CU (320 -> 1280: 80 KB, 576 -> 96: 72 KB) is staged in phases as well: the blocks' weight
    // staging, K loops and epilogues overlap across more resident blocks (SNNHIP_CONV_1X1_PHASE_KB: the phase budget, default 26, 0 = off)
    if (!f16 && M >= 8192.0) {
        const char* pk = snnhip::option("SNNHIP_CONV_1X1_PHASE_KB");
        const size_t budget = static_cast<size_t>(pk ? atoi(pk) : 26) * 1024; // (c4 at batch 256: off 3.42 ms, 53 KB 3.37, 40 KB 3.36, 26 / 16 / 12 KB 3.34)
        const size_t wBytes = static_cast<size_t>(g.IC) * 32 * NT * 4;
        if (budget && wBytes > budget) KP = std::max(KP, static_cast<int>(std::min<size_t>(8, (wBytes + budget - 1) / budget)));
    }
    if (g.IC >= 128 && static_cast<long long>((nTiles + 3) / 4) * ((g.OC + 32 * NT - 1) / (32 * NT)) < cus / 4 && !(sw && atoi(sw) == 2)) return SNNHIP_E_UNSUPPORTED;
    const int BN = 32 * NT, ocBlocks = (g.OC + BN - 1) / BN, OCp = ocBlocks * BN, nChunks = f16 ? (g.IC + 15) / 16 : g.IC / 8;
    const int phaseChunks = KP == 1 ? nChunks : ((nChunks + KP - 1) / KP + 7) / 8 * 8; // a multiple of the deepest prefetch (8 chunks)

    auto* plan = new Conv1x1StreamPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC);
    plan->epi4 = epi4;
    plan->p = StreamParams{static_cast<int>(M), g.IC, g.OC, nChunks, phaseChunks, nTiles, g.useBN, g.sh, g.W, g.H * g.W, g.OW, g.OH * g.OW, nullptr,
                           make_act_cfg(g.addAct >= 0 ? g.addAct : SNNHIP_ACT_NONE, g.addLeaky)};
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->fusedAdd = g.addAct >= 0;
    if (plan->fusedAdd) plan->numInputs = 2;
    plan->ldsBytes = static_cast<size_t>(std::min(nChunks, phaseChunks)) * 2 * BN * 16 + (f16 ? static_cast<size_t>(4) * 32 * (BN + 8) * 2 : 0);
    const bool simple = act_is_simple(g.act);
    if (f16) plan->kernel = NT == 3 ? pick16<3>(simple) : NT == 2 ? pick16<2>(simple) : pick16<1>(simple);
    // eight tiles per block for the fp32 layers that widen at least threefold (see the kernel), SNNHIP_CONV_1X1_WAVES=4|8 pins it
    int waves = (!f16 && g.OC >= 3 * g.IC) ? 8 : 4;
    if (const char* wv = snnhip::option("SNNHIP_CONV_1X1_WAVES"))
        if (!f16 && (atoi(wv) == 4 || atoi(wv) == 8)) waves = atoi(wv);
    plan->waves = waves;
    if (!f16) plan->kernel = NT == 3 ? pick<3>(simple, waves) : NT == 2 ? pick<2>(simple, waves) : pick<1>(simple, waves);
    int gx = (nTiles + waves - 1) / waves; // one 32-pixel tile per wave
    plan->grid = dim3(gx, ocBlocks);
    // ---- the widening fp32 layers as persistent blocks with a loader wave (conv1x1_march_kernel): whole 96-channel blocks, stride 1, no fused Add, a weight
    // slice of at most 64 KB, and enough tiles that every CU's block runs several steps.  SNNHIP_CONV_1X1_MARCH=0 keeps the kernel above, =1 forces it on small layers
    bool march = false;
    int mTS = 0, mPI = 0;
    {
        const char* mo = snnhip::option("SNNHIP_CONV_1X1_MARCH");
        const int mode = mo ? atoi(mo) : -1;
        const int pieces = (32 * (g.IC / 4 + 1) + 63) / 64;
        const bool shape = !f16 && NT == 3 && g.sh == 1 && !plan->fusedAdd && g.OC % 96 == 0 && static_cast<size_t>(g.IC) * 96 * 4 <= 64 * 1024 && pieces <= 21;
        const bool worth = g.OC >= 3 * g.IC && static_cast<long long>(nTiles) * ocBlocks >= 8LL * cus;
        if (shape && mode != 0 && (worth || mode == 1)) {
            auto fn = pick_march(pieces, !simple ? 0 : g.act == SNNHIP_ACT_LEAKY ? 1 : 2, &mTS, &mPI);
            const size_t lds = static_cast<size_t>(nChunks) * 2 * BN * 16 + static_cast<size_t>(2) * mTS * mPI * 64 * 16;
            if (fn && lds <= 160 * 1024) {
                march = true;
                plan->kernel = fn;
                plan->ldsBytes = lds;
                plan->waves = 3 * mTS + 1;
                int segs = std::max(1, cus / ocBlocks);
                plan->p.segTiles = (nTiles + segs - 1) / segs;
                segs = (nTiles + plan->p.segTiles - 1) / plan->p.segTiles;
                plan->p.phaseChunks = nChunks; // the whole slice is resident
                gx = segs;
                plan->grid = dim3(segs, ocBlocks);
            }
        }
    }
    if (plan->ldsBytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(plan->kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(plan->ldsBytes)) != hipSuccess) {
        delete plan;
        return SNNHIP_E_UNSUPPORTED;
    }

    // Wp[ocBlock][chunk][h][BN] x 16 bytes: fp32 component j = w[oc][ic = 8*chunk + 4*h + j]; fp16 half j = w[oc][ic = 16*chunk + 8*h + j] (zero past IC)
    std::vector<float> wp(static_cast<size_t>(ocBlocks) * nChunks * 2 * BN * 4, 0.0f);
    _Float16* wph = reinterpret_cast<_Float16*>(wp.data());
    const int per = f16 ? 8 : 4;
    for (int ob = 0; ob < ocBlocks; ++ob)
        for (int c = 0; c < nChunks; ++c)
            for (int hh = 0; hh < 2; ++hh)
                for (int o = 0; o < BN; ++o) {
                    const int oc = ob * BN + o;
                    if (oc >= g.OC) continue;
                    const size_t slot = ((static_cast<size_t>(ob) * nChunks + c) * 2 + hh) * BN + o;
                    for (int j = 0; j < per; ++j) {
                        const int ic = (c * 2 + hh) * per + j;
                        if (ic >= g.IC) continue;
                        const float wv = w_oihw[static_cast<size_t>(oc) * g.IC + ic];
                        if (f16) wph[slot * 8 + j] = static_cast<_Float16>(wv);
                        else wp[slot * 4 + j] = wv;
                    }
                }
    std::vector<float> epiP(static_cast<size_t>(OCp) * 4, 0.0f);
    std::memcpy(epiP.data(), epi4.data(), sizeof(float) * 4 * static_cast<size_t>(g.OC));
    int rc = plan->upload(wp.data(), wp.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epiP.data(), epiP.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = g.N; plan->inDims[1] = g.H; plan->inDims[2] = g.W; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->dtype = g.dtype;
    plan->flops = 2.0 * g.IC * g.OC * M;
    const double esz = f16 ? 2.0 : 4.0;
    plan->bytes = esz * (static_cast<double>(g.N) * g.H * g.W * g.IC + M * g.OC + static_cast<double>(g.OC) * g.IC); // same accounting as the general kernel
    char buf[256];
    snprintf(buf, sizeof(buf), "conv2d_mfma_%s k=1x1 s=%d ic=%d oc=%d stream: wave = 32px x %doc, %d waves per block, grid %dx%d, lds=%zuB", f16 ? "f16_32x32x16" : "f32_32x32x2", g.sh, g.IC, g.OC,
             BN, waves, gx, ocBlocks, plan->ldsBytes);
    plan->desc = buf;
    if (march) {
        snprintf(buf, sizeof(buf), "conv2d_mfma_f32_32x32x2 k=1x1 s=1 ic=%d oc=%d march: persistent blocks (grid %dx%d, %d tiles each, weights resident), loader wave "
                 "(LDS-DMA %d KB/tile) + %d compute waves, %d tiles/step, lds=%zuB kernel=conv1x1_march_kernel<%d,%d>",
                 g.IC, g.OC, gx, ocBlocks, plan->p.segTiles, mPI, 3 * mTS, mTS, plan->ldsBytes, mTS, mPI);
        plan->desc = buf;
    }
    if (KP > 1 && !march) plan->desc += " (weight slice in " + std::to_string((nChunks + phaseChunks - 1) / phaseChunks) + " K phases)";
    if (plan->fusedAdd) {
        plan->desc += " +add";
        plan->bytes += esz * M * g.OC;
    }
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
