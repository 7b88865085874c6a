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
    const int gx = (nTiles + waves - 1) / waves; // one 32-pixel tile per wave
    plan->grid = dim3(gx, ocBlocks);
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
    if (KP > 1) plan->desc += " (weight slice in " + std::to_string((nChunks + phaseChunks - 1) / phaseChunks) + " K phases)";
    if (plan->fusedAdd) {
        plan->desc += " +add";
        plan->bytes += esz * M * g.OC;
    }
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
