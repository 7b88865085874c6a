// espcn_fused.hip -- chain fusion for the ESPCN-shaped part of the hot path (BASELINE config 2), fp32.
//
// The reference runs one compute dispatch + one full barrier per layer (core/src/ic2/vulkanRenderpass.cpp:257-259)
// and round-trips every intermediate through a texture.  Here a linear chain of plans is rewritten into two kernels:
//
//   kernel A  conv KxK (IC=1 -> 16, K in {3,5}) + act  ->  conv 3x3 (16 -> 16) + act       [fp32 MFMA, LDS-resident mid tensor]
//             replaces two passes of shadertemplate_vk_conv2d.comp:148-347
//   kernel B  conv 3x3 (16 -> 4) + act  ->  depth-to-space(2) + tanh                         [VALU, HBM-bound]
//             replaces shadertemplate_vk_conv2d.comp + shadertemplate_vk_subpixel.comp:43-71
//
// Kernel A, per 64x8 output tile (256 threads = 4 waves, 46 KB LDS, 3 blocks/CU = __launch_bounds__(256, 3)):
//   phase 0  input halo tile (70x14, 1 channel) -> LDS, zero outside the image (constant padding)
//   phase 1  conv1 on the 66x10 halo region as a GEMM  D[oc][px] = W1[oc][tap] * im2col[tap][px]  with
//            v_mfma_f32_16x16x4_f32 (K = 25 taps padded to 28), bias/BN/act fused, zeroed outside the image (it is
//            conv2's zero padding), written to LDS as [row][col][16ch] with a 16-byte-slot XOR swizzle
//   phase 2  conv2 as 9 taps x 4 MFMAs per 16-pixel group: B operand = one ds_read_b128 (4 input channels of one
//            pixel), A operand = weights held in 36 VGPRs for the whole kernel; 8 groups (=accumulators) per wave
//   epilogue bias/BN/act, 16-byte stores: one wave store = 16 pixels x 64 B contiguous NHWC
// fp32 MFMA is bit-for-bit an fp32 fma chain (no reduced precision), so the 1e-4 parity bound holds as for VALU code.
#include <hip/hip_ext.h>

#include <cstdlib>

#include "epilogue.h"
#include "snnhip_internal.h"

// developer hook: tools/tune_espcn.hip defines SNNHIP_STAMP(k) to record s_memtime per wave and phase
#ifndef SNNHIP_STAMP
#define SNNHIP_STAMP(k)
#endif

namespace snnhip {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// XCD-aware tile order (guide T1): workgroup b runs on XCD b % 8 and every XCD has a private 4 MiB L2.  Handing each XCD
// a contiguous run of tiles keeps the halo rows/columns that neighbouring tiles share inside one L2 instead of
// re-fetching them through the fabric.  Bijective for any grid size; purely a performance hint.
__device__ __forceinline__ int xcd_tile_order(int b, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

struct FusedAParams {
    int N, H, W, tilesX, tilesY;
    ActCfg act1, act2;
};

// K1 = first conv's kernel size (3 or 5), pad = K1/2.  TW multiple of 16, TH multiple of 4.
template <int K1, int TW, int TH, bool SIMPLE, int U = 3, int WPS = 3>
__global__ __launch_bounds__(256, WPS) void conv_kxk_c1o16_conv3x3_c16o16_kernel(FusedAParams p, const float* __restrict__ x,
                                                                            const float* __restrict__ wA1, const float* __restrict__ wA2,
                                                                            const float* __restrict__ ep1, const float* __restrict__ ep2,
                                                                            float* __restrict__ y) {
    constexpr int P1 = K1 / 2;
    constexpr int C1W = TW + 2, C1H = TH + 2;                 // conv1 output region needed by conv2 (halo 1)
    constexpr int INW = TW + 2 + 2 * P1, INH = TH + 2 + 2 * P1; // input region needed by conv1 on that region
    constexpr int KS1 = (K1 * K1 + 3) / 4;                    // MFMA K-steps of conv1
    constexpr int NG1 = (C1H * C1W + 15) / 16;                // 16-pixel groups of phase 1
    constexpr int GPR = TW / 16;                              // groups per output row
    constexpr int RPW = TH / 4;                               // output rows per wave
    constexpr int G = GPR * RPW;                              // groups (= accumulators) per wave in phase 2

    __shared__ __attribute__((aligned(16))) float smem[C1H * C1W * 16 + INH * INW];
    float* s_c1 = smem;
    float* s_in = smem + C1H * C1W * 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int px = lane & 15, g = lane >> 4;

    int b = xcd_tile_order(blockIdx.x, gridDim.x);
    const int tx = b % p.tilesX;
    b /= p.tilesX;
    const int ty = b % p.tilesY;
    const int n = b / p.tilesY;
    const int x0 = tx * TW, y0 = ty * TH;
    const float* xn = x + static_cast<size_t>(n) * p.H * p.W;

    SNNHIP_STAMP(0);
    // ---- phase 0: input tile (origin y0-1-P1, x0-1-P1), zero padded
    {
        constexpr int NLD = (INH * INW + 255) / 256;
        float v[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) { // all loads in flight before the first use
            const int idx = tid + k * 256;
            const int r = idx / INW, c = idx - r * INW;
            const int gy = y0 - 1 - P1 + r, gx = x0 - 1 - P1 + c;
            v[k] = 0.0f;
            if (idx < INH * INW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) v[k] = xn[static_cast<size_t>(gy) * p.W + gx];
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            if (tid + k * 256 < INH * INW) s_in[tid + k * 256] = v[k];
    }

    // ---- weights -> registers (host packed them in lane order)
    float a1[KS1];
#pragma unroll
    for (int s = 0; s < KS1; ++s) a1[s] = wA1[s * 64 + lane];
    float a2[36];
#pragma unroll
    for (int t = 0; t < 36; ++t) a2[t] = wA2[t * 64 + lane];
    float sc1[4], sh1[4], sc2[4], sh2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sc1[r] = ep1[(4 * g + r) * 2];
        sh1[r] = ep1[(4 * g + r) * 2 + 1];
        sc2[r] = ep2[(4 * g + r) * 2];
        sh2[r] = ep2[(4 * g + r) * 2 + 1];
    }
    // this lane's tap offsets for conv1: K-step s covers taps 4s..4s+3, lane group g supplies tap 4s+g
    int off1[KS1];
    bool tapok[KS1];
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
        int t = 4 * s + g;
        tapok[s] = t < K1 * K1;
        if (!tapok[s]) t = K1 * K1 - 1;
        off1[s] = (t / K1) * INW + (t % K1);
    }
    SNNHIP_STAMP(1);
    __syncthreads();
    SNNHIP_STAMP(2);

    // ---- phase 1: conv1 over the C1H x C1W halo region, pixels flattened into 16-wide groups.
    // U groups per iteration = U independent MFMA accumulation chains (one chain alone is bound by the 40-cycle
    // dependent-accumulator latency of v_mfma_f32_16x16x4_f32).
    for (int grp0 = wv * U; grp0 < NG1; grp0 += 4 * U) {
        f32x4 acc[U];
        const float* src[U];
        int rr[U], cc[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pi = (grp0 + u) * 16 + px;
            valid[u] = pi < C1H * C1W;
            const int pc = valid[u] ? pi : C1H * C1W - 1;
            rr[u] = pc / C1W;
            cc[u] = pc - rr[u] * C1W;
            src[u] = s_in + rr[u] * INW + cc[u];
            acc[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            float bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                bv[u] = src[u][off1[s]];
                if (!tapok[s]) bv[u] = 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bv[u], acc[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int gy = y0 - 1 + rr[u], gx = x0 - 1 + cc[u];
            const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            float4 o;
            o.x = inside ? apply_act<SIMPLE>(p.act1, fmaf(acc[u][0], sc1[0], sh1[0]), 0.0f) : 0.0f;
            o.y = inside ? apply_act<SIMPLE>(p.act1, fmaf(acc[u][1], sc1[1], sh1[1]), 0.0f) : 0.0f;
            o.z = inside ? apply_act<SIMPLE>(p.act1, fmaf(acc[u][2], sc1[2], sh1[2]), 0.0f) : 0.0f;
            o.w = inside ? apply_act<SIMPLE>(p.act1, fmaf(acc[u][3], sc1[3], sh1[3]), 0.0f) : 0.0f;
            if (valid[u]) {
                const int slot = g ^ (((cc[u] >> 2) & 1) << 1); // conflict-free ds_read_b128 in phase 2 for every tap shift
                *reinterpret_cast<float4*>(s_c1 + (rr[u] * C1W + cc[u]) * 16 + slot * 4) = o;
            }
        }
    }
    SNNHIP_STAMP(3);
    __syncthreads();
    SNNHIP_STAMP(4);

    // ---- phase 2: conv2, G accumulators per wave
    f32x4 acc2[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) acc2[gi] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int fy = tap / 3, fx = tap % 3;
        float4 bv[G];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const int row = wv * RPW + gi / GPR, col0 = (gi % GPR) * 16;
            const int cc = col0 + px + fx;
            const int slot = g ^ (((cc >> 2) & 1) << 1);
            bv[gi] = *reinterpret_cast<const float4*>(s_c1 + ((row + fy) * C1W + cc) * 16 + slot * 4);
        }
#pragma unroll
        for (int gi = 0; gi < G; ++gi) acc2[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[tap * 4 + 0], bv[gi].x, acc2[gi], 0, 0, 0);
#pragma unroll
        for (int gi = 0; gi < G; ++gi) acc2[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[tap * 4 + 1], bv[gi].y, acc2[gi], 0, 0, 0);
#pragma unroll
        for (int gi = 0; gi < G; ++gi) acc2[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[tap * 4 + 2], bv[gi].z, acc2[gi], 0, 0, 0);
#pragma unroll
        for (int gi = 0; gi < G; ++gi) acc2[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[tap * 4 + 3], bv[gi].w, acc2[gi], 0, 0, 0);
    }

    SNNHIP_STAMP(5);
    // ---- epilogue: lane holds output channels 4g..4g+3 of pixel (row, col0+px)
    float* yn = y + static_cast<size_t>(n) * p.H * p.W * 16;
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        const int gy = y0 + wv * RPW + gi / GPR, gx = x0 + (gi % GPR) * 16 + px;
        if (gy < p.H && gx < p.W) {
            float4 o;
            o.x = apply_act<SIMPLE>(p.act2, fmaf(acc2[gi][0], sc2[0], sh2[0]), 0.0f);
            o.y = apply_act<SIMPLE>(p.act2, fmaf(acc2[gi][1], sc2[1], sh2[1]), 0.0f);
            o.z = apply_act<SIMPLE>(p.act2, fmaf(acc2[gi][2], sc2[2], sh2[2]), 0.0f);
            o.w = apply_act<SIMPLE>(p.act2, fmaf(acc2[gi][3], sc2[3], sh2[3]), 0.0f);
            *reinterpret_cast<float4*>(yn + (static_cast<size_t>(gy) * p.W + gx) * 16 + g * 4) = o;
        }
    }
    SNNHIP_STAMP(6);
}

// Kernel A, Winograd variant (default): same fusion, but conv2 (3x3, 16 -> 16) is evaluated as F(2x2, 3x3):
//     Y = At [ (G g Gt) .* (Bt d B) ] A      per 2x2 output block and (oc, ic) pair            (Lavin & Gray 2016)
// which needs 16 multiplies per 4 outputs instead of 36 -> 2.25x fewer MFMA flops for the layer that owns 70 % of the
// network's arithmetic.  All transform coefficients are 0, +-1 (input/output) or +-1/2 (weights, done on the host in
// double precision), so the fp32 result differs from the direct sum by a few ulp (parity tests: <= 1e-4 as before).
//   block   = 32x16 output pixels = 16x8 Winograd tiles; 256 threads = 4 waves; 58.6 KB LDS -> 2 blocks/CU
//   phase 1 = conv1 (MFMA, as in the direct kernel) into LDS [row][col parity][col/2][16ch]; the column de-interleave makes
//             the stride-2 tile reads of phase 2 consecutive, the slot XOR (bit 2 of the linear pixel index) makes them
//             conflict-free for the ds_read_b128 lane groups
//   phase 2 = per wave 2 groups of 16 tiles (one tile row each): lane (tile t = lane%16, g = lane/16) loads its 4x4 input
//             patch for channels 4g..4g+3 (16 ds_read_b128), then per transform column nu: input transform (VALU, float4
//             adds), 16 MFMAs  M[xi][oc][tile] += U[xi,nu][oc][ic] * V[xi,nu][ic][tile]  (v_mfma_f32_16x16x4_f32, U read
//             from LDS as one b128 per position), output transform folded into the 2x2x4 result registers
//   epilogue= bias/BN/act, four 16-byte stores per lane
struct WinoTile {
    static constexpr int TW = 32;
};

// conv1 K-step -> tap assignment of the Winograd kernel: K-step s, lane group g (= MFMA k index) handle
//   s <  K1        : tap (row g, col s)            valid iff g < K1          -> LDS address = base + g*INW + s   (s is an immediate)
//   s == K1 + j    : tap (row 4, col 4j + g)       valid iff K1 == 5, col < 5 -> LDS address = base + 4*INW + g + 4j
// (invalid (s, g) carry a zero weight and read an initialised location).  Returns the tap index or -1.
inline int wino_conv1_tap(int K1, int s, int g) {
    if (s < K1) return g < K1 ? g * K1 + s : -1;
    const int col = 4 * (s - K1) + g;
    return (K1 > 4 && col < K1) ? 4 * K1 + col : -1;
}
constexpr int wino_conv1_ksteps(int K1) { return K1 + (K1 > 4 ? 2 : 0); }

// AM: 0 = any activation (run-time switch), 1 = cheap family (branch-free med3 form), 2 = both layers ReLU
template <int AM>
__device__ __forceinline__ float act_mode(const ActCfg& a, float v) {
    if (AM == 2) return fmaxf(v, 0.0f);
    if (AM == 1) return apply_act<true>(a, v, 0.0f);
    return epi_act(a.act, a.leaky, v, 0.0f);
}

template <int K1, int TH, int AM, int WPS>
__global__ __launch_bounds__(256, WPS) void conv_kxk_c1o16_wino3x3_c16o16_kernel(FusedAParams p, const float* __restrict__ x,
                                                                           const float* __restrict__ wA1, const float* __restrict__ wU,
                                                                           const float* __restrict__ ep1, const float* __restrict__ ep2,
                                                                           float* __restrict__ y) {
    constexpr int TW = WinoTile::TW, U = 2;
    constexpr int P1 = K1 / 2;
    constexpr int C1W = TW + 2, C1H = TH + 2;
    constexpr int C1P = 40, HALFP = 20;                        // LDS row pitch / odd-column plane offset, in pixels (see phase 2)
    constexpr int INW = TW + 2 + 2 * P1, INH = TH + 2 + 2 * P1;
    constexpr int INP = 48;                                    // LDS row pitch of the input tile: == 16 (mod 32) puts the four tap rows a
                                                               // wave reads in one ds_read_b32 (lane group g -> row g) on disjoint banks
    constexpr int KS1 = wino_conv1_ksteps(K1);
    constexpr int NG1 = (C1H * C1W + 15) / 16;                 // 16-pixel groups of phase 1
    constexpr int GPW = (NG1 + 3) / 4;                         // groups per wave (contiguous range)
    constexpr int NIT = (GPW + U - 1) / U;
    constexpr int GROUPS2 = TH / 8;                            // Winograd tile rows (= phase-2 groups) per wave
    constexpr int NLD = (INH * INP + 8 + 255) / 256;
    static_assert(INW <= INP, "input pitch");
    static_assert(C1W / 2 + 1 <= HALFP && HALFP + C1W / 2 <= C1P, "plane layout");

    __shared__ __attribute__((aligned(16))) float smem[C1H * C1P * 16 + 4096 + INH * INP + 8];
    float* s_c1 = smem;
    float* s_U = smem + C1H * C1P * 16;
    float* s_in = s_U + 4096;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int px = lane & 15, g = lane >> 4;
    const int ntiles = p.tilesX * p.tilesY * p.N;

    // Persistent blocks (grid = WPS per CU): weights / epilogue constants are loaded once per block and the input tile of the
    // NEXT tile is fetched into NLD registers while this tile is computed, so no wave ever waits on HBM inside the loop.
    auto tile_origin = [&](int t, int& n, int& x0, int& y0) {
        int b = xcd_tile_order(t, ntiles);
        const int tx = b % p.tilesX;
        b /= p.tilesX;
        const int ty = b % p.tilesY;
        n = b / p.tilesY;
        x0 = tx * TW;
        y0 = ty * TH;
    };
    float vin[NLD];
    auto issue_loads = [&](int t) { // input tile (origin y0-1-P1, x0-1-P1), zero padded (+8 zero floats: invalid taps read them)
        int n, x0, y0;
        tile_origin(t, n, x0, y0);
        const float* xn = x + static_cast<size_t>(n) * p.H * p.W;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + k * 256;
            const int r = idx / INP, c = idx - r * INP;
            const int gy = y0 - 1 - P1 + r, gx = x0 - 1 - P1 + c;
            vin[k] = 0.0f;
            if (r < INH && c < INW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) vin[k] = xn[static_cast<size_t>(gy) * p.W + gx];
        }
    };
    auto store_input = [&]() {
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            if (tid + k * 256 < INH * INP + 8) s_in[tid + k * 256] = vin[k];
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    SNNHIP_STAMP(0);
    issue_loads(tile);
    {
        float4 u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = reinterpret_cast<const float4*>(wU)[tid + k * 256];
#pragma unroll
        for (int k = 0; k < 4; ++k) reinterpret_cast<float4*>(s_U)[tid + k * 256] = u[k];
    }
    store_input();

    // 5x5: the 7th K step would carry ONE tap (row 4, column 4) in a 4-deep MFMA -- 32 pipe cycles for 64 useful FMAs per lane group.  That tap goes
    // to the VALU instead: every lane reads the input value under its own pixel and adds w[oc][24] * x to its four accumulators (4 FMAs, ~10 cycles)
    #ifdef SNNHIP_ESPCN_TAP25_MFMA // experiment builds (tools/exp_one.sh): the round-3 form, all 7 steps on the matrix pipe
    constexpr bool kValuTap = false;
#else
    constexpr bool kValuTap = K1 == 5;
#endif
    constexpr int KSM = kValuTap ? KS1 - 1 : KS1; // K steps on the matrix pipe
    float a1[KSM];
#pragma unroll
    for (int s = 0; s < KSM; ++s) a1[s] = wA1[s * 64 + lane];
    float sc1[4], sh1[4], sc2[4], sh2[4], w24[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sc1[r] = ep1[(4 * g + r) * 2];
        sh1[r] = ep1[(4 * g + r) * 2 + 1];
        sc2[r] = ep2[(4 * g + r) * 2];
        sh2[r] = ep2[(4 * g + r) * 2 + 1];
        w24[r] = kValuTap ? wA1[(KS1 - 1) * 64 + 4 * g + r] : 0.0f; // step 6 holds w[oc][24] at lane oc (its lane group 0)
    }
    const int rowTap = (g < K1 ? g : 0) * INP; // K-steps s < K1: tap row g (invalid g: zero weight, any initialised row)
    const int lastTap = 4 * INP + g;           // K-steps s >= K1 (K1 == 5): tap row 4, col g (+4)
    SNNHIP_STAMP(1);
    __syncthreads();
    SNNHIP_STAMP(2);

  for (;;) {
    int n, x0, y0;
    tile_origin(tile, n, x0, y0);
    const int next = tile + gridDim.x;
    const bool more = next < ntiles;
    // The two blocks of a CU (b and b + grid / 2: workgroups fill every CU's first slot before any second one) take turns at the higher wave priority,
    // tile by tile -- conv2d_widep_f16.hip's rule, measured here as well: kernel A 75.4 -> 73.5 us in six of six ABAB pairs on one box (round 5), the
    // headline 9.26 k -> 9.41 k images/s there.  Recorded as measured, not derived
#ifndef SNNHIP_ESPCN_NO_PRIO_ALT // (experiment builds switch it off)
    if (((tile / static_cast<int>(gridDim.x)) + (blockIdx.x >= (gridDim.x >> 1) ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(0);
#endif
    if (more) issue_loads(next);
    const bool border = x0 == 0 || y0 == 0 || x0 + TW >= p.W || y0 + TH >= p.H; // wave-uniform

    // ---- phase 1: conv1 over the C1H x C1W region, pixels flattened into 16-wide groups; wave wv owns groups
    // [wv*GPW, wv*GPW+GPW), two per iteration (two independent MFMA chains).  The LDS operands of iteration it+1 are
    // fetched before the MFMAs of iteration it (the loop is fully unrolled, so this is register renaming, not copies).
    {
        float bv[2][U][KS1]; // (5x5: slot KS1 - 1 holds the input value of the VALU tap)
        int rr[2][U], cc[2][U];
        bool valid[2][U];
        auto fetch = [&](int it, int buf) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int grp = wv * GPW + it * U + u;
                const int pi = grp * 16 + px;
                valid[buf][u] = (it * U + u < GPW) && pi < C1H * C1W;
                const int pc = valid[buf][u] ? pi : 0;
                rr[buf][u] = pc / C1W;
                cc[buf][u] = pc - rr[buf][u] * C1W;
                const float* src = s_in + rr[buf][u] * INP + cc[buf][u];
                const float* srcRow = src + rowTap;
#pragma unroll
                for (int s = 0; s < KSM; ++s) bv[buf][u][s] = s < K1 ? srcRow[s] : src[lastTap + 4 * (s - K1)];
                if (kValuTap) bv[buf][u][KS1 - 1] = src[4 * INP + 4]; // tap (4, 4) under this lane's pixel, whatever its lane group
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int cur = it & 1;
            if (it + 1 < NIT) fetch(it + 1, cur ^ 1);
            f32x4 acc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0], bv[cur][u][0], f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < KSM; ++s)
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bv[cur][u][s], acc[u], 0, 0, 0);
            if (kValuTap) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[u][k] = fmaf(w24[k], bv[cur][u][KS1 - 1], acc[u][k]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float4 o;
                o.x = act_mode<AM>(p.act1, fmaf(acc[u][0], sc1[0], sh1[0]));
                o.y = act_mode<AM>(p.act1, fmaf(acc[u][1], sc1[1], sh1[1]));
                o.z = act_mode<AM>(p.act1, fmaf(acc[u][2], sc1[2], sh1[2]));
                o.w = act_mode<AM>(p.act1, fmaf(acc[u][3], sc1[3], sh1[3]));
                if (border) { // only tiles on the image border have conv1 pixels outside the image: they are conv2's zero padding
                    const int gy = y0 - 1 + rr[cur][u], gx = x0 - 1 + cc[cur][u];
                    if (!(gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)) o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
                if (valid[cur][u]) {
                    const int c = cc[cur][u];
                    const int pl = rr[cur][u] * C1P + (c & 1) * HALFP + (c >> 1);
                    const int slot = g ^ (((pl >> 2) & 1) << 1);
                    *reinterpret_cast<float4*>(s_c1 + pl * 16 + slot * 4) = o;
                }
            }
        }
    }
    SNNHIP_STAMP(3);
    __syncthreads(); // c1 complete; every wave is done reading s_in
    if (more) store_input();
    SNNHIP_STAMP(4);

    // ---- phase 2: Winograd conv2.  Lane = (tile column t = px, channel quad g).
    // c1 pixel (r, c) lives at linear pixel pl = r*C1P + (c&1)*HALFP + (c>>1), 16-byte slot  q ^ 2*((pl>>2)&1).
    // Patch element (i, j) of tile (trow, t): pl = (2 trow + i)*C1P + (j&1)*HALFP + t + (j>>1).  C1P = 40 leaves bit 2 of
    // pl alone, HALFP = 20 flips it, so the swizzle term is one of two per-lane values and everything else is an
    // immediate offset: 4 address registers serve all 16 patch loads.
    const float* uBase = s_U + (px * 4 + (g ^ (((px >> 2) & 1) << 1))) * 4; // + pos*256: U[pos][oc = px][ic = 4g..4g+3]
    const float* dBase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = px + (j >> 1);
        const int slot = g ^ ((((t >> 2) & 1) ^ (j & 1)) << 1);
        dBase[j] = s_c1 + (wv * (2 * GROUPS2) * C1P + (j & 1) * HALFP + t) * 16 + slot * 4;
    }
    float* yn = y + static_cast<size_t>(n) * p.H * p.W * 16;
#pragma unroll
    for (int gi = 0; gi < GROUPS2; ++gi) {
        const int trow = wv * GROUPS2 + gi; // tile row: output rows 2*trow, 2*trow+1; patch rows 2*trow .. 2*trow+3 of the c1 tile
        f32x4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x4*>(dBase[j] + (2 * gi + i) * C1P * 16);
        f32x4 Y[2][2];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            // (d B)[:, nu], then V[xi] = (Bt (dB))[xi]
            f32x4 e[4], V[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                e[i] = nu == 0 ? d[i][0] - d[i][2] : nu == 1 ? d[i][1] + d[i][2] : nu == 2 ? d[i][2] - d[i][1] : d[i][1] - d[i][3];
            V[0] = e[0] - e[2];
            V[1] = e[1] + e[2];
            V[2] = e[2] - e[1];
            V[3] = e[1] - e[3];
            f32x4 u4[4], m[4];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) u4[xi] = *reinterpret_cast<const f32x4*>(uBase + (xi * 4 + nu) * 256);
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) m[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(u4[xi][0], V[xi][0], f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
#pragma unroll
            for (int kk = 1; kk < 4; ++kk)
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) m[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(u4[xi][kk], V[xi][kk], m[xi], 0, 0, 0);
            // output transform: T[a] = (At M)[a][nu];  Y[a][b] += T[a] * At[b][nu]
            const f32x4 T0 = m[0] + m[1] + m[2];
            const f32x4 T1 = m[1] - m[2] - m[3];
            if (nu == 0) {
                Y[0][0] = T0;
                Y[1][0] = T1;
            } else if (nu == 1) {
                Y[0][0] += T0;
                Y[1][0] += T1;
                Y[0][1] = T0;
                Y[1][1] = T1;
            } else if (nu == 2) {
                Y[0][0] += T0;
                Y[1][0] += T1;
                Y[0][1] -= T0;
                Y[1][1] -= T1;
            } else {
                Y[0][1] -= T0;
                Y[1][1] -= T1;
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int gy = y0 + 2 * trow + a, gx = x0 + 2 * px + bb;
                if (!border || (gy < p.H && gx < p.W)) {
                    float4 o;
                    o.x = act_mode<AM>(p.act2, fmaf(Y[a][bb][0], sc2[0], sh2[0]));
                    o.y = act_mode<AM>(p.act2, fmaf(Y[a][bb][1], sc2[1], sh2[1]));
                    o.z = act_mode<AM>(p.act2, fmaf(Y[a][bb][2], sc2[2], sh2[2]));
                    o.w = act_mode<AM>(p.act2, fmaf(Y[a][bb][3], sc2[3], sh2[3]));
                    *reinterpret_cast<float4*>(yn + (static_cast<size_t>(gy) * p.W + gx) * 16 + g * 4) = o;
                }
            }
    }
    SNNHIP_STAMP(5);
    if (!more) break;
    __syncthreads(); // c1 consumed, next input tile visible
    tile = next;
  }
    SNNHIP_STAMP(6);
}

struct FusedBParams {
    int N, H, W, tilesX, tilesY;
    ActCfg act;
    unsigned magicX, magicY; // ceil(2^32 / tilesX), ceil(2^32 / tilesY): tile decode without integer division (exact for block ids < 2^16 * ...)
};

// conv 3x3 (16 -> 4, zero padding 1) + act, then depth-to-space(2) + tanh.  One thread = one input-resolution pixel
// = a 2x2 block of the output image.  LDS tile [TH+2][TW+2] pixels, 64 B each, 16-byte slots XOR-swizzled (conflict-free
// b128 reads for 64 consecutive pixels, 21.8 KB per block -> 7 blocks/CU); weights are wave-uniform => scalar loads, FMAs take them as SGPR operands.
template <int TW, int TH, bool SIMPLE>
__global__ __launch_bounds__(256) void conv3x3_c16o4_d2s_tanh_kernel(FusedBParams p, const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ ep, float* __restrict__ y) {
    // LDS tile as four channel-quad PLANES, s_x[q][pixel] float4: a wave's 64 pixels (2 rows x 32) read 512 contiguous bytes per row from one
    // plane -- conflict-free without a swizzle -- and every operand address of the tap loop is ONE per-thread base + a wave-uniform tap offset + a
    // compile-time plane offset.  (The kernel is VALU-issue bound: rocprofv3 counted 708 VALU instructions per wave of which 288 are the
    // packed FMAs; the previous [pixel][quad ^ swizzle] layout spent 21 VALU instructions per tap on addresses, this one 2.)
    constexpr int TWH = TW + 2, THH = TH + 2, PLANE = THH * TWH * 4; // floats per plane
    static_assert(TW * TH == 256, "one thread per pixel");
    __shared__ __attribute__((aligned(16))) float s_x[4 * PLANE];

    const int tid = threadIdx.x;
    // tile decode on the scalar unit: the divisions by tilesX / tilesY are mul-hi by host-computed magic numbers (a run-time integer division of
    // a uniform value still compiles to ~20 VALU instructions of float reciprocal arithmetic, and this kernel is VALU-issue bound)
    const unsigned bid = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(xcd_tile_order(blockIdx.x, gridDim.x)));
    const unsigned bq = p.tilesX == 1 ? bid : __umulhi(bid, p.magicX);
    const int tx = static_cast<int>(bid - bq * p.tilesX);
    const unsigned n_u = p.tilesY == 1 ? bq : __umulhi(bq, p.magicY);
    const int ty = static_cast<int>(bq - n_u * p.tilesY), n = static_cast<int>(n_u);
    const int x0 = tx * TW, y0 = ty * TH;
    const float* xn = x + static_cast<size_t>(n) * p.H * p.W * 16;

    {
        const bool interior = x0 >= 1 && y0 >= 1 && x0 + TW + 1 <= p.W && y0 + TH + 1 <= p.H; // block-uniform: 95 % of the tiles at 1080p
        if (interior) {
            // no bounds tests, no zero fill, no index arithmetic: thread t < 4*TWH owns float4 t of EVERY halo row (a row of the tile is 4*TWH
            // contiguous float4 in memory), so its THH loads are one pointer walked by the image pitch and its THH LDS stores one offset walked by
            // the tile pitch.  The other threads (the fourth wave entirely) skip the staging: fewer instructions issued in total is what counts.
            if (tid < 4 * TWH) {
                const float* src = xn + (static_cast<size_t>(y0 - 1) * p.W + (x0 - 1)) * 16 + tid * 4;
                float4 rowv[THH];
#pragma unroll
                for (int rr = 0; rr < THH; ++rr) rowv[rr] = *reinterpret_cast<const float4*>(src + static_cast<size_t>(rr) * p.W * 16);
                float* dst = s_x + (tid & 3) * PLANE + (tid >> 2) * 4;
#pragma unroll
                for (int rr = 0; rr < THH; ++rr) *reinterpret_cast<float4*>(dst + rr * TWH * 4) = rowv[rr];
            }
        } else {
            constexpr int NLD = (THH * TWH * 4 + 255) / 256;
            float4 v[NLD];
#pragma unroll
            for (int k = 0; k < NLD; ++k) { // every load of the halo tile is in flight before the first LDS write
                const int idx = tid + k * 256;
                const int q = idx & 3, pix = idx >> 2;
                const int r = pix / TWH, c = pix - r * TWH;
                const int gy = y0 - 1 + r, gx = x0 - 1 + c;
                v[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (idx < THH * TWH * 4 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                    v[k] = *reinterpret_cast<const float4*>(xn + (static_cast<size_t>(gy) * p.W + gx) * 16 + q * 4);
            }
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int idx = tid + k * 256;
                if (idx < THH * TWH * 4) *reinterpret_cast<float4*>(s_x + (idx & 3) * PLANE + (idx >> 2) * 4) = v[k];
            }
        }
    }
    __syncthreads();

    const int c = tid % TW, r = tid / TW;
    // Packed fp32 FMAs: a wave64 v_fma_f32 occupies the VALU for 4 cycles on this kernel (measured: 20.3 M VALU instructions
    // = 20.6 M quad-cycles busy), v_pk_fma_f32 retires two FMAs per lane in the same slot.  The accumulators are kept as two
    // float2 so that every FMA is a v_pk_fma_f32 with the weight pair in an SGPR pair.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc01 = {0.0f, 0.0f}, acc23 = {0.0f, 0.0f};
    const float* base = s_x + (r * TWH + c) * 4;
    // one tap (64 uniform weights = 64 SGPRs) per iteration: unrolling further only spills SGPRs.  (Prefetching tap t+1's operand quads from
    // LDS does not pay: LDS and scalar loads share lgkmcnt, so the wait for the next weights also waits for the prefetch.)
#pragma unroll 1
    for (int fy = 0; fy < 3; ++fy) {
#pragma unroll 1
        for (int fx = 0; fx < 3; ++fx) {
            const int tap = fy * 3 + fx;
            const float* src = base + (fy * TWH + fx) * 4; // wave-uniform offset
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 xv = *reinterpret_cast<const float4*>(src + q * PLANE); // compile-time plane offset -> ds_read_b128 offset:
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float* wr = w + (tap * 16 + q * 4 + i) * 4; // uniform address -> s_load
                    const f32x2 xx = {xs[i], xs[i]};
                    const f32x2 w01 = {wr[0], wr[1]}, w23 = {wr[2], wr[3]};
                    acc01 = __builtin_elementwise_fma(xx, w01, acc01);
                    acc23 = __builtin_elementwise_fma(xx, w23, acc23);
                }
            }
        }
    }
    const float acc[4] = {acc01.x, acc01.y, acc23.x, acc23.y};
    const int gy = y0 + r, gx = x0 + c;
    if (gy < p.H && gx < p.W) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fast_tanh(apply_act<SIMPLE>(p.act, fmaf(acc[k], ep[2 * k], ep[2 * k + 1]), 0.0f));
        float* yn = y + static_cast<size_t>(n) * (2 * p.H) * (2 * p.W);
        // channel 2*dy+dx -> output pixel (2y+dy, 2x+dx)  (depth_to_space, fs_subpixel.glsl:41-64)
        *reinterpret_cast<float2*>(yn + static_cast<size_t>(2 * gy) * (2 * p.W) + 2 * gx) = make_float2(o[0], o[1]);
        *reinterpret_cast<float2*>(yn + static_cast<size_t>(2 * gy + 1) * (2 * p.W) + 2 * gx) = make_float2(o[2], o[3]);
    }
}

// Kernel B, Winograd variant (default): conv 3x3 (16 -> 4) as F(2x2, 3x3) on the matrix cores, then depth-to-space(2) + tanh.
// With only 4 output channels a 16-wide MFMA tile would be 3/4 padding; v_mfma_f32_4x4x1_16B_f32 instead runs 16
// independent 4x4x1 outer products per instruction: block = 4 Winograd tiles, rows = the 4 output channels, one input
// channel per instruction.  Lane l is Winograd tile l of its wave (64 tiles = 32 x 2), so both transforms are lane-local:
//   per channel quad q: 16 ds_read_b128 (the tile's 4x4 input patch), Bt d B (32 float4 adds), then per transform
//   position 4 MFMAs  M[pos][oc][tile] += U[pos][oc][ic] * V[pos][ic][tile]   (U: one broadcast ds_read_b128 per position)
//   after the 4 quads: At M A (24 float4 adds), bias/BN/act, tanh, and the 4x4 block of output pixels the tile maps to
//   under depth-to-space is written as four 16-byte stores.
// 256 MFMAs (8 cycles each) + ~800 VALU per 256 pixels instead of 9216 scalar FMAs.  fp32 MFMA executes on the same FMA
// lanes as VALU code (tools/ubench_issue.hip: their issue times add up), so the win is the 2.25x cut in multiplies plus the
// removal of the per-tap scalar weight loads.  Block = 64 x 16 pixels, 80 KB LDS -> 2 blocks/CU.
template <bool SIMPLE>
__global__ __launch_bounds__(256, 2) void conv3x3_c16o4_wino_d2s_tanh_kernel(FusedBParams p, const float* __restrict__ x, const float* __restrict__ wU,
                                                                        const float* __restrict__ ep, float* __restrict__ y) {
    constexpr int TW = 64, TH = 16, XW = TW + 2, XH = TH + 2, HALF = XW / 2;
    constexpr int NPIX = XH * XW, NLD = (NPIX + 255) / 256; // halo pixels per tile; float4 per thread and channel quad
    __shared__ __attribute__((aligned(16))) float s_x[NPIX * 16];
    __shared__ __attribute__((aligned(16))) float s_U[1024];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ntiles = p.tilesX * p.tilesY * p.N;

    // Persistent blocks (grid = 2 per CU) with a quad-granular software pipeline: the tile is consumed one channel quad
    // (16-byte slot of every pixel) at a time, so as soon as all waves are done with quad q of this tile, quad q of the NEXT
    // tile -- fetched into 5 float4 registers per thread while quad q was being computed -- overwrites it.  The HBM/MALL stream
    // and the MFMA/VALU work overlap inside every block; without this all blocks of a launch load together and compute
    // together (measured: 44 % of the block lifetime in the load phase).
    // Halo tile layout: pixel (r, c) at linear index pl = r*XW + (c&1)*HALF + (c>>1) (columns de-interleaved so that the
    // stride-2 patch reads of 32 adjacent tiles are consecutive), 16-byte slot q ^ ((pl>>2)&3) (conflict-free ds_read_b128).
    auto tile_origin = [&](int t, int& n, int& x0, int& y0) {
        int b = xcd_tile_order(t, ntiles);
        const int tx = b % p.tilesX;
        b /= p.tilesX;
        const int ty = b % p.tilesY;
        n = b / p.tilesY;
        x0 = tx * TW;
        y0 = ty * TH;
    };
    // tile-independent staging descriptors of this thread's halo pixels
    int rc[NLD], ldst[NLD]; // (r << 8) | c   and   byte offset of slot bits 0 of the pixel, swizzle term folded in
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int pix = tid + k * 256;
        const int pp = pix < NPIX ? pix : 0;
        const int r = pp / XW, c = pp - r * XW;
        const int pl = r * XW + (c & 1) * HALF + (c >> 1);
        rc[k] = pix < NPIX ? ((r << 8) | c) : -1;
        ldst[k] = pl * 64 + (((pl >> 2) & 3) << 4);
    }
    float4 v[NLD];
    const float* xt = nullptr; // image base of the tile being fetched
    int fx0 = 0, fy0 = 0;
    auto begin_fetch = [&](int t) {
        int n;
        tile_origin(t, n, fx0, fy0);
        xt = x + static_cast<size_t>(n) * p.H * p.W * 16;
    };
    auto issue_loads = [&](int q) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int gy = fy0 - 1 + (rc[k] >> 8), gx = fx0 - 1 + (rc[k] & 255);
            v[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (rc[k] >= 0 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                v[k] = *reinterpret_cast<const float4*>(xt + (static_cast<size_t>(gy) * p.W + gx) * 16 + q * 4);
        }
    };
    auto store_lds = [&](int q) {
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            if (rc[k] >= 0) *reinterpret_cast<float4*>(reinterpret_cast<char*>(s_x) + (ldst[k] ^ (q << 4))) = v[k];
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    SNNHIP_STAMP(0);
    begin_fetch(tile);
    reinterpret_cast<float4*>(s_U)[tid] = reinterpret_cast<const float4*>(wU)[tid];
    { // first tile: all four quads in flight at once
        float4 v0[4][NLD];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            issue_loads(q);
#pragma unroll
            for (int k = 0; k < NLD; ++k) v0[q][k] = v[k];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) v[k] = v0[q][k];
            store_lds(q);
        }
    }
    SNNHIP_STAMP(1);
    __syncthreads();
    SNNHIP_STAMP(2);

    float sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc[k] = ep[2 * k];
        sh[k] = ep[2 * k + 1];
    }
    const int tcol = lane & 31, trow = 2 * wv + (lane >> 5); // this lane's Winograd tile: output pixels (2 trow + a, 2 tcol + b)
    int off[4][4];                                             // byte offset of patch pixel (i, j), slot bits of quad 0
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pl = (2 * trow + i) * XW + (j & 1) * HALF + tcol + (j >> 1);
            off[i][j] = pl * 64 + (((pl >> 2) & 3) << 4);
        }
    const char* sxb = reinterpret_cast<const char*>(s_x);
    const float* uLane = s_U + (lane & 3) * 4; // + (pos*4 + q)*16 floats: U[pos][oc = lane&3][ic = 4q .. 4q+3]

    for (;;) {
        int n, x0, y0;
        tile_origin(tile, n, x0, y0);
        const int next = tile + gridDim.x;
        const bool more = next < ntiles;
        if (more) begin_fetch(next);

        f32x4 M[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (more) issue_loads(q);
            f32x4 d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x4*>(sxb + (off[i][j] ^ (q << 4)));
                // V = Bt d B, in place: columns first (d B), then rows
#ifndef SNNHIP_BW_SKIP_TRANSFORM
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 e0 = d[i][0] - d[i][2], e1 = d[i][1] + d[i][2], e2 = d[i][2] - d[i][1], e3 = d[i][1] - d[i][3];
                d[i][0] = e0;
                d[i][1] = e1;
                d[i][2] = e2;
                d[i][3] = e3;
            }
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                const f32x4 v0 = d[0][nu] - d[2][nu], v1 = d[1][nu] + d[2][nu], v2 = d[2][nu] - d[1][nu], v3 = d[1][nu] - d[3][nu];
                d[0][nu] = v0;
                d[1][nu] = v1;
                d[2][nu] = v2;
                d[3][nu] = v3;
            }
#endif
            // 16 positions x 4 channels; 4 positions in flight so consecutive MFMAs never share an accumulator
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                f32x4 u4[4];
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) u4[nu] = *reinterpret_cast<const f32x4*>(uLane + ((xi * 4 + nu) * 4 + q) * 16);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int nu = 0; nu < 4; ++nu) {
                        const int pos = xi * 4 + nu;
                        if (q == 0 && k == 0)
                            M[pos] = __builtin_amdgcn_mfma_f32_4x4x1f32(u4[nu][k], d[xi][nu][k], f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
                        else
                            M[pos] = __builtin_amdgcn_mfma_f32_4x4x1f32(u4[nu][k], d[xi][nu][k], M[pos], 0, 0, 0);
                    }
            }
            if (more) {
                __syncthreads(); // every wave has consumed quad q of this tile
                store_lds(q);
            }
        }
        SNNHIP_STAMP(3);

        // ---- Y = At M A  (per output channel = accumulator register)
        f32x4 Y[2][2];
        {
            f32x4 T[2][4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                T[0][nu] = M[0 * 4 + nu] + M[1 * 4 + nu] + M[2 * 4 + nu];
                T[1][nu] = M[1 * 4 + nu] - M[2 * 4 + nu] - M[3 * 4 + nu];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                Y[a][0] = T[a][0] + T[a][1] + T[a][2];
                Y[a][1] = T[a][1] - T[a][2] - T[a][3];
            }
        }
        SNNHIP_STAMP(4);
        // ---- epilogue: bias/BN/act, tanh; channel 2*dy+dx of input pixel (yy, xx) -> output pixel (2yy+dy, 2xx+dx)
        // (depth_to_space, fs_subpixel.glsl:41-64): the tile's 2x2 pixels x 4 channels are a 4x4 block of the output image
        float* yn = y + static_cast<size_t>(n) * (2 * p.H) * (2 * p.W);
        const int gy0 = y0 + 2 * trow, gx0 = x0 + 2 * tcol;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                float o[4];
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int oc = 2 * dy + dx;
                        o[2 * bb + dx] = fast_tanh(apply_act<SIMPLE>(p.act, fmaf(Y[a][bb][oc], sc[oc], sh[oc]), 0.0f));
                    }
                const int gy = gy0 + a;
                float* row = yn + static_cast<size_t>(2 * gy + dy) * (2 * p.W) + 2 * gx0;
                if (gy < p.H) {
                    if (gx0 + 1 < p.W) {
                        *reinterpret_cast<float4*>(row) = make_float4(o[0], o[1], o[2], o[3]);
                    } else if (gx0 < p.W) {
                        *reinterpret_cast<float2*>(row) = make_float2(o[0], o[1]);
                    }
                }
            }
        SNNHIP_STAMP(5);
        if (!more) break;
        __syncthreads(); // the last quad of the next tile is in place
        tile = next;
    }
    SNNHIP_STAMP(6);
}

// (scale, shift) per channel so that epilogue = act(acc*scale + shift):  scale = bnScale, shift = bnScale*(bias-mean)+beta
std::vector<float> fold_epilogue(const std::vector<float>& epi4, int OC, int useBN) {
    std::vector<float> out(static_cast<size_t>(OC) * 2);
    for (int o = 0; o < OC; ++o) {
        const float bias = epi4[o * 4 + 0], sc = epi4[o * 4 + 1], mean = epi4[o * 4 + 2], beta = epi4[o * 4 + 3];
        out[o * 2 + 0] = useBN ? sc : 1.0f;
        out[o * 2 + 1] = useBN ? sc * (bias - mean) + beta : bias;
    }
    return out;
}

bool plain_act(int act) { return act >= 0 && act <= SNNHIP_ACT_SILU; }

bool is_same_conv(const ConvGeom& g, int k, int ic, int oc) { // the ESPCN kernels are fp32
    return g.dtype == SNNHIP_F32 && g.preMode == 0 && g.kh == k && g.kw == k && g.IC == ic && g.OC == oc && g.sh == 1 && g.sw == 1 && g.padx == k / 2 && g.pady == k / 2 &&
           (g.padMode == SNNHIP_PAD_CONSTANT || g.padMode == SNNHIP_PAD_NONE) && g.OH == g.H && g.OW == g.W && plain_act(g.act);
}

constexpr int A_TW = 64, A_TH = 8;
constexpr int W_TH = 16, W_WPS = 2; // Winograd kernel A: tile 32 x W_TH, W_WPS blocks (waves/SIMD) per CU
constexpr int B_TW = 32, B_TH = 8;

// Chain rule F: Conv2D -> InstanceNorm.  The convolution (conv2d_mfma, fp16 LDS epilogue) leaves (mean, M2) of every output tile and channel
// next to its output; the InstanceNorm's statistics sweep -- one of its three passes over the tensor -- is replaced by a fold over those
// tile records, and its normalise pass runs in place on the convolution's output.  Both plans are borrowed (the chain or the caller owns them).
struct ConvInstanceNormPlan : snnhip_plan {
    snnhip_plan* conv = nullptr; // the convolution, or the InstanceNorm -> convolution of rule I that wraps it
    snnhip_plan* norm = nullptr;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        int rc = conv->invoke(in, nIn, out);
        if (rc != SNNHIP_OK) return rc;
        return instancenorm_apply_tile_stats(norm, tiles, out);
    }
    TileStatsRef tiles;
};

// Graph rule I: InstanceNorm -> [UpSampling] -> [Pad] -> Conv2D.  The norm runs its statistics sweep and fold only; the convolution (a copy
// the chain owns, built with ConvGeom::normShift / normMul) reads the norm's INPUT and normalises while it stages -- the normalised tensor is never
// written or re-read.  The norm plan is borrowed: its parameters and statistics buffers are the ones the convolution was given.
struct InstanceNormConvPlan : snnhip_plan {
    snnhip_plan* norm = nullptr;
    snnhip_plan* conv = nullptr;
    TileStatsRef tiles; // rule F in front: the convolution that PRODUCED in[0] left tile statistics -- a fold over them instead of the sweep
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        const int rc = instancenorm_run_stats(norm, in[0], &tiles);
        return rc != SNNHIP_OK ? rc : conv->invoke(in, nIn, out);
    }
};

struct ChainPlan : snnhip_plan {
    enum Kind { PLAIN, FUSED_A, FUSED_B, FUSED_S };
    struct Step {
        Kind kind = PLAIN;
        snnhip_plan* plain = nullptr; // borrowed
        FusedAParams a{};
        FusedBParams b{};
        int k1 = 5;
        bool wino = false; // FUSED_A / FUSED_B: the 3x3 conv as Winograd F(2x2,3x3) (default) or direct (SNNHIP_ESPCN_A / _B = direct)
        float *w1 = nullptr, *w2 = nullptr, *e1 = nullptr, *e2 = nullptr, *w3 = nullptr, *e3 = nullptr;
        alignas(8) char streamCfg[kStreamCfgBytes] = {};
        int outDims[4] = {0, 0, 0, 0};
        std::string desc;
        double flops = 0, bytes = 0; // algorithmic work of this launch (fused steps: inputs once + outputs once + weights)
    };
    std::vector<Step> steps;
    std::vector<snnhip_tensor*> mids; // owned intermediates between steps
    std::vector<snnhip_plan*> owned;  // plans built by the chain itself (rule D: a convolution with the Pad layer folded into its staging)

    ~ChainPlan() override {
        for (auto* t : mids) snnhip_tensor_free(t);
        for (auto* q : owned) delete q;
    }
    int numSteps() const override { return static_cast<int>(steps.size()); }
    std::string stepDesc(int i) const override { return steps[i].desc; }
    void stepCost(int i, double* f, double* b) const override {
        *f = steps[i].flops;
        *b = steps[i].bytes;
    }
    bool profilesItself() const override { return true; }

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == numInputs, "chain: expects %d input(s), got %d", numInputs, nIn);
        const snnhip_tensor* src = in[0];
        SNNHIP_REQUIRE(src->n == inDims[0] && src->h == inDims[1] && src->w == inDims[2] && src->c == inDims[3],
                       "chain: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", src->n, src->h, src->w, src->c, inDims[0], inDims[1], inDims[2], inDims[3]);
        SNNHIP_REQUIRE(out->n == outDims[0] && out->h == outDims[1] && out->w == outDims[2] && out->c == outDims[3],
                       "chain: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n, out->h, out->w, out->c, outDims[0], outDims[1], outDims[2],
                       outDims[3]);
        for (size_t i = 0; i < steps.size(); ++i) {
            Step& s = steps[i];
            snnhip_tensor* dst = (i + 1 == steps.size()) ? out : mids[i];
            hipEvent_t evStart = nullptr, evStop = nullptr;
            TraceScope traceScope(s.desc, s.flops, s.bytes); // a PLAIN step's plan opens its own scope inside this one
            if (profiling) {
                int rc = (s.kind == FUSED_A || s.kind == FUSED_B) ? profAcquire(static_cast<int>(i), &evStart, &evStop) : profBegin(static_cast<int>(i));
                if (rc != SNNHIP_OK) return rc;
            }
            if (s.kind == PLAIN) {
                // a chain with two inputs: the second one belongs to its LAST step (InstanceNorm -> Add behind a run of layers, rules F + H)
                const snnhip_tensor* two[2] = {src, nIn > 1 ? in[1] : nullptr};
                int rc = s.plain->invoke(two, (i + 1 == steps.size()) ? nIn : 1, dst);
                if (rc != SNNHIP_OK) return rc;
            } else if (s.kind == FUSED_S) {
                int rc = espcn_stream_launch(ctx->stream, s.streamCfg, src->data, s.w1, s.e1, s.w2, s.e2, s.w3, s.e3, dst->data);
                if (rc != SNNHIP_OK) return rc;
            } else if (s.kind == FUSED_A && s.wino) {
                const int ntilesA = s.a.tilesX * s.a.tilesY * s.a.N;
                const int slotsA = W_WPS * (ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256);
                dim3 grid(ntilesA < slotsA ? ntilesA : slotsA); // persistent: W_WPS blocks per CU walk the tile list
                const bool simple = act_is_simple(s.a.act1.act) && act_is_simple(s.a.act2.act);
#define SNNHIP_LAUNCH_W(K, AM)                                                                                                                        \
    SNNHIP_LAUNCH_EV((conv_kxk_c1o16_wino3x3_c16o16_kernel<K, W_TH, AM, W_WPS>), grid, dim3(256), 0, ctx->stream, evStart, evStop, s.a, src->data, \
                          s.w1, s.w2, s.e1, s.e2, dst->data)
                const int am = (s.a.act1.act == SNNHIP_ACT_RELU && s.a.act2.act == SNNHIP_ACT_RELU) ? 2 : (simple ? 1 : 0);
                if (s.k1 == 5) {
                    if (am == 2) SNNHIP_LAUNCH_W(5, 2); else if (am == 1) SNNHIP_LAUNCH_W(5, 1); else SNNHIP_LAUNCH_W(5, 0);
                } else {
                    if (am == 2) SNNHIP_LAUNCH_W(3, 2); else if (am == 1) SNNHIP_LAUNCH_W(3, 1); else SNNHIP_LAUNCH_W(3, 0);
                }
#undef SNNHIP_LAUNCH_W
                SNNHIP_CHECK_HIP(hipGetLastError());
            } else if (s.kind == FUSED_A) {
                dim3 grid(s.a.tilesX * s.a.tilesY * s.a.N);
                const bool simple = act_is_simple(s.a.act1.act) && act_is_simple(s.a.act2.act);
#define SNNHIP_LAUNCH_A(K, S)                                                                                                                         \
    SNNHIP_LAUNCH_EV((conv_kxk_c1o16_conv3x3_c16o16_kernel<K, A_TW, A_TH, S>), grid, dim3(256), 0, ctx->stream, evStart, evStop, s.a, src->data, \
                          s.w1, s.w2, s.e1, s.e2, dst->data)
                if (s.k1 == 5) {
                    if (simple) SNNHIP_LAUNCH_A(5, true); else SNNHIP_LAUNCH_A(5, false);
                } else {
                    if (simple) SNNHIP_LAUNCH_A(3, true); else SNNHIP_LAUNCH_A(3, false);
                }
#undef SNNHIP_LAUNCH_A
                SNNHIP_CHECK_HIP(hipGetLastError());
            } else if (s.wino) {
                // one block per tile
                const int ntiles = s.b.tilesX * s.b.tilesY * s.b.N;
                dim3 grid(ntiles);
                if (act_is_simple(s.b.act.act)) {
                    SNNHIP_LAUNCH_EV((conv3x3_c16o4_wino_d2s_tanh_kernel<true>), grid, dim3(256), 0, ctx->stream, evStart, evStop, s.b, src->data,
                                          s.w1, s.e1, dst->data);
                } else {
                    SNNHIP_LAUNCH_EV((conv3x3_c16o4_wino_d2s_tanh_kernel<false>), grid, dim3(256), 0, ctx->stream, evStart, evStop, s.b, src->data,
                                          s.w1, s.e1, dst->data);
                }
                SNNHIP_CHECK_HIP(hipGetLastError());
            } else {
                dim3 grid(s.b.tilesX * s.b.tilesY * s.b.N);
                if (act_is_simple(s.b.act.act)) {
                    SNNHIP_LAUNCH_EV((conv3x3_c16o4_d2s_tanh_kernel<B_TW, B_TH, true>), grid, dim3(256), 0, ctx->stream, evStart, evStop, s.b,
                                          src->data, s.w1, s.e1, dst->data);
                } else {
                    SNNHIP_LAUNCH_EV((conv3x3_c16o4_d2s_tanh_kernel<B_TW, B_TH, false>), grid, dim3(256), 0, ctx->stream, evStart, evStop, s.b,
                                          src->data, s.w1, s.e1, dst->data);
                }
                SNNHIP_CHECK_HIP(hipGetLastError());
            }
            if (profiling && !(s.kind == FUSED_A || s.kind == FUSED_B)) {
                int rc = profEnd(static_cast<int>(i));
                if (rc != SNNHIP_OK) return rc;
            }
            src = dst;
        }
        return SNNHIP_OK;
    }
};

} // namespace

int make_chain_plan(snnhip_ctx* ctx, snnhip_plan* const* plans, int n, snnhip_plan** out) {
    // Default = the two-kernel fusion (rules A+B).  SNNHIP_ESPCN_FUSION=stream selects the single row-streaming kernel
    // (rule C, espcn_stream.hip): parity-tested, 20 B/px of HBM traffic, but measured slower on MI355X so far
    // (206 us vs 123+45 us per 1080p frame) because its per-wave dependency chain starves the matrix pipe.
    // ---- rule E: Conv2D (MFMA kernel) + Add -> one launch, the residual is added in the convolution's epilogue.  The fused plan takes TWO
    // inputs, snnhip_plan_run_n(plan, {conv input, residual}, 2, out), so it is returned as is instead of being wrapped into a ChainPlan.
    if (n == 2 && !snnhip::option("SNNHIP_NO_ADD_FUSION")) {
        auto* cv = dynamic_cast<ConvPlanBase*>(plans[0]);
        auto* ad = dynamic_cast<EltwisePlanBase*>(plans[1]);
        if (cv && ad && ad->mode == 0 && !cv->depthwise && cv->g.addAct < 0 && cv->desc.rfind("conv2d_mfma", 0) == 0 && cv->g.act != SNNHIP_ACT_SILU_QUIRK &&
            ad->d.N == cv->g.N && ad->d.H == cv->g.OH && ad->d.W == cv->g.OW && ad->d.C == cv->g.OC) {
            ConvGeom g2 = cv->g;
            g2.addAct = ad->d.act;
            g2.addLeaky = ad->d.leaky;
            return make_conv2d_mfma_plan(ctx, g2, cv->w_oihw.data(), cv->epi4, out);
        }
    }
    for (int i = 0; i < n; ++i)
        if (plans[i]->numInputs != 1 && !(i == n - 1 && plans[i]->numInputs == 2)) {
            set_error("chain fusion: plan %d takes %d inputs (only the last plan of a chain may take two)", i, plans[i]->numInputs);
            return SNNHIP_E_UNSUPPORTED;
        }
    const char* mode = snnhip::option("SNNHIP_ESPCN_FUSION");
    const bool allowStream = mode && strcmp(mode, "stream") == 0;
    auto* chain = new ChainPlan();
    chain->ctx = ctx;
    chain->numInputs = plans[n - 1]->numInputs;
    memcpy(chain->inDims, plans[0]->inDims, sizeof(chain->inDims));
    memcpy(chain->outDims, plans[n - 1]->outDims, sizeof(chain->outDims));
    int fusedCount = 0;
    int rc = SNNHIP_OK;
    for (int i = 0; i < n && rc == SNNHIP_OK;) {
        ChainPlan::Step st;
        auto* c0 = dynamic_cast<ConvPlanBase*>(plans[i]);
        auto* c1 = (i + 1 < n) ? dynamic_cast<ConvPlanBase*>(plans[i + 1]) : nullptr;
        auto* sp1 = (i + 1 < n) ? dynamic_cast<SubpixelPlanBase*>(plans[i + 1]) : nullptr;
        // the chain must be shape-consistent
        // (a dense layer consumes any [N,H,W,C] tensor flattened in HWC order: same batch, same element count)
        auto count3 = [](const int* d) { return static_cast<long long>(d[1]) * d[2] * d[3]; };
        if (i + 1 < n && (plans[i]->outDims[0] != plans[i + 1]->inDims[0] || count3(plans[i]->outDims) != count3(plans[i + 1]->inDims))) {
            set_error("chain: plan %d output %dx%dx%dx%d does not feed plan %d input %dx%dx%dx%d", i, plans[i]->outDims[0], plans[i]->outDims[1],
                      plans[i]->outDims[2], plans[i]->outDims[3], i + 1, plans[i + 1]->inDims[0], plans[i + 1]->inDims[1], plans[i + 1]->inDims[2],
                      plans[i + 1]->inDims[3]);
            rc = SNNHIP_E_INVALID;
            break;
        }
        auto* c2 = (i + 2 < n) ? dynamic_cast<ConvPlanBase*>(plans[i + 2]) : nullptr;
        auto* sp3 = (i + 3 < n) ? dynamic_cast<SubpixelPlanBase*>(plans[i + 3]) : nullptr;
        const bool pairA = c0 && c1 && !c0->depthwise && !c1->depthwise && (is_same_conv(c0->g, 5, 1, 16) || is_same_conv(c0->g, 3, 1, 16)) &&
                           is_same_conv(c1->g, 3, 16, 16);
        if (allowStream && pairA && c2 && sp3 && !c2->depthwise && is_same_conv(c2->g, 3, 16, 4) && sp3->d.factor == 2 &&
            sp3->d.mode == SNNHIP_SUBPIXEL_D2S && sp3->d.C == 4 && memcmp(plans[i + 1]->outDims, plans[i + 2]->inDims, sizeof(int) * 4) == 0 &&
            memcmp(plans[i + 2]->outDims, plans[i + 3]->inDims, sizeof(int) * 4) == 0) {
            // ---- rule C: the whole ESPCN pattern as one row-streaming kernel (espcn_stream.hip)
            const ConvGeom& g0 = c0->g;
            const int K1 = g0.kh, taps1 = K1 * K1;
            st.kind = ChainPlan::FUSED_S;
            st.k1 = K1;
            static_assert(sizeof(st.streamCfg) >= 1, "");
            if (espcn_stream_step_size() > sizeof(st.streamCfg)) {
                set_error("internal: stream cfg blob too small");
                rc = SNNHIP_E_INVALID;
                break;
            }
            espcn_stream_configure(st.streamCfg, g0.N, g0.H, g0.W, K1, g0.act, g0.leaky, c1->g.act, c1->g.leaky, c2->g.act, c2->g.leaky,
                                   ctx->props.multiProcessorCount);
            const int ks1 = (taps1 + 3) / 4;
            std::vector<float> w1s(static_cast<size_t>(ks1) * 64, 0.0f), wA2(36 * 64), w3s(9 * 16 * 4);
            for (int s = 0; s < ks1; ++s)
                for (int l = 0; l < 64; ++l) {
                    const int oc = l & 15, t = 4 * s + (l >> 4);
                    if (t < taps1) w1s[s * 64 + l] = c0->w_oihw[static_cast<size_t>(oc) * taps1 + t];
                }
            for (int tap = 0; tap < 9; ++tap)
                for (int j = 0; j < 4; ++j)
                    for (int l = 0; l < 64; ++l) {
                        const int oc = l & 15, ic = 4 * (l >> 4) + j;
                        wA2[(tap * 4 + j) * 64 + l] = c1->w_oihw[(static_cast<size_t>(oc) * 16 + ic) * 9 + tap];
                    }
            for (int dx = 0; dx < 3; ++dx) // w3r[((dx*4+q)*4+i)*12 + dy*4 + o] = W3[o][ic = 4q+i][dy][dx]
                for (int ic = 0; ic < 16; ++ic)
                    for (int dy = 0; dy < 3; ++dy)
                        for (int o = 0; o < 4; ++o) w3s[(dx * 16 + ic) * 12 + dy * 4 + o] = c2->w_oihw[(static_cast<size_t>(o) * 16 + ic) * 9 + dy * 3 + dx];
            std::vector<float> e1 = fold_epilogue(c0->epi4, 16, g0.useBN), e2 = fold_epilogue(c1->epi4, 16, c1->g.useBN),
                               e3 = fold_epilogue(c2->epi4, 4, c2->g.useBN);
            rc = chain->upload(w1s.data(), w1s.size(), &st.w1);
            if (rc == SNNHIP_OK) rc = chain->upload(wA2.data(), wA2.size(), &st.w2);
            if (rc == SNNHIP_OK) rc = chain->upload(w3s.data(), w3s.size(), &st.w3);
            if (rc == SNNHIP_OK) rc = chain->upload(e1.data(), e1.size(), &st.e1);
            if (rc == SNNHIP_OK) rc = chain->upload(e2.data(), e2.size(), &st.e2);
            if (rc == SNNHIP_OK) rc = chain->upload(e3.data(), e3.size(), &st.e3);
            memcpy(st.outDims, sp3->outDims, sizeof(st.outDims));
            char buf[320];
            espcn_stream_describe(st.streamCfg, buf, sizeof(buf));
            st.desc = buf;
            st.flops = c0->flops + c1->flops + c2->flops;
            st.bytes = 4.0 * (static_cast<double>(g0.N) * g0.H * g0.W * (1 + 4) + 16.0 * taps1 + 16.0 * 16 * 9 + 4.0 * 16 * 9);
            i += 4;
            ++fusedCount;
        } else if (pairA) {
            // ---- rule A
            const ConvGeom& g0 = c0->g;
            const int K1 = g0.kh, taps1 = K1 * K1, ks1 = (taps1 + 3) / 4;
            st.kind = ChainPlan::FUSED_A;
            st.k1 = K1;
            const char* amode = snnhip::option("SNNHIP_ESPCN_A");
            st.wino = !(amode && strcmp(amode, "direct") == 0);
            const int aTW = st.wino ? WinoTile::TW : A_TW, aTH = st.wino ? W_TH : A_TH;
            st.a = FusedAParams{g0.N, g0.H, g0.W, up_div(g0.W, aTW), up_div(g0.H, aTH), make_act_cfg(g0.act, g0.leaky), make_act_cfg(c1->g.act, c1->g.leaky)};
            std::vector<float> wA1(static_cast<size_t>(ks1) * 64, 0.0f), wA2(36 * 64);
            for (int s = 0; s < ks1; ++s)
                for (int l = 0; l < 64; ++l) {
                    const int oc = l & 15, t = 4 * s + (l >> 4);
                    if (t < taps1) wA1[s * 64 + l] = c0->w_oihw[static_cast<size_t>(oc) * taps1 + t];
                }
            if (st.wino) {
                wA1.assign(static_cast<size_t>(wino_conv1_ksteps(K1)) * 64, 0.0f);
                for (int s = 0; s < wino_conv1_ksteps(K1); ++s)
                    for (int l = 0; l < 64; ++l) {
                        const int t = wino_conv1_tap(K1, s, l >> 4);
                        if (t >= 0) wA1[s * 64 + l] = c0->w_oihw[static_cast<size_t>(l & 15) * taps1 + t];
                    }
                // U[pos = xi*4+nu][oc][ic] = (G g Gt)[xi][nu], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], in double, stored as the
                // kernel's LDS image: float4 index (pos*16 + oc)*4 + (q ^ 2*((oc>>2)&1)) holds ic = 4q .. 4q+3
                static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
                wA2.assign(4096, 0.0f);
                for (int oc = 0; oc < 16; ++oc)
                    for (int ic = 0; ic < 16; ++ic) {
                        const float* gk = &c1->w_oihw[(static_cast<size_t>(oc) * 16 + ic) * 9];
                        double tmp[4][3];
                        for (int xi = 0; xi < 4; ++xi)
                            for (int v = 0; v < 3; ++v) tmp[xi][v] = Gm[xi][0] * gk[0 * 3 + v] + Gm[xi][1] * gk[1 * 3 + v] + Gm[xi][2] * gk[2 * 3 + v];
                        for (int xi = 0; xi < 4; ++xi)
                            for (int nu = 0; nu < 4; ++nu) {
                                const double u = tmp[xi][0] * Gm[nu][0] + tmp[xi][1] * Gm[nu][1] + tmp[xi][2] * Gm[nu][2];
                                const int q = ic >> 2, slot = q ^ (((oc >> 2) & 1) << 1);
                                wA2[(((xi * 4 + nu) * 16 + oc) * 4 + slot) * 4 + (ic & 3)] = static_cast<float>(u);
                            }
                    }
            } else {
                for (int tap = 0; tap < 9; ++tap)
                    for (int j = 0; j < 4; ++j)
                        for (int l = 0; l < 64; ++l) {
                            const int oc = l & 15, ic = 4 * (l >> 4) + j;
                            wA2[(tap * 4 + j) * 64 + l] = c1->w_oihw[(static_cast<size_t>(oc) * 16 + ic) * 9 + tap];
                        }
            }
            std::vector<float> e1 = fold_epilogue(c0->epi4, 16, g0.useBN), e2 = fold_epilogue(c1->epi4, 16, c1->g.useBN);
            rc = chain->upload(wA1.data(), wA1.size(), &st.w1);
            if (rc == SNNHIP_OK) rc = chain->upload(wA2.data(), wA2.size(), &st.w2);
            if (rc == SNNHIP_OK) rc = chain->upload(e1.data(), e1.size(), &st.e1);
            if (rc == SNNHIP_OK) rc = chain->upload(e2.data(), e2.size(), &st.e2);
            memcpy(st.outDims, c1->outDims, sizeof(st.outDims));
            char buf[320];
            // MFMA flops actually issued (2048 per v_mfma_f32_16x16x4_f32): conv1 on the halo region with K padded to a multiple
            // of 4, conv2 either direct (36 per 16 pixels) or Winograd (64 per 16 tiles = 64 pixels)
            double mfmaFlops;
            {
                const double tiles = static_cast<double>(st.a.tilesX) * st.a.tilesY * g0.N;
                const int c1px = (aTW + 2) * (aTH + 2);
                const double conv1 = st.wino ? 4.0 * (((((c1px + 15) / 16) + 3) / 4 + 1) / 2 * 2) * (wino_conv1_ksteps(K1) - (K1 == 5 ? 1 : 0)) : ((c1px + 15) / 16) * ks1; // (5x5: the 25th tap runs on the VALU)
                const double conv2 = st.wino ? (aTW / 2) * (aTH / 2) / 16 * 64.0 : aTW * aTH / 16 * 36.0;
                mfmaFlops = tiles * (conv1 + conv2) * 2048.0;
            }
            snprintf(buf, sizeof(buf), "fused[conv%dx%d(1->16)+conv3x3(16->16)%s] mfma_f32_16x16x4 tile=%dx%d kernel=%s mfma_flops=%.6g", K1, K1,
                     st.wino ? " winograd F(2x2,3x3)" : "", aTW, aTH,
                     st.wino ? "conv_kxk_c1o16_wino3x3_c16o16_kernel" : "conv_kxk_c1o16_conv3x3_c16o16_kernel", mfmaFlops);
            st.desc = buf;
            st.flops = c0->flops + c1->flops;
            st.bytes = 4.0 * (static_cast<double>(g0.N) * g0.H * g0.W * (1 + 16) + 16.0 * taps1 + 16.0 * 16 * 9);
            i += 2;
            ++fusedCount;
        } else if (c0 && sp1 && !c0->depthwise && is_same_conv(c0->g, 3, 16, 4) && sp1->d.factor == 2 && sp1->d.mode == SNNHIP_SUBPIXEL_D2S &&
                   sp1->d.C == 4) {
            // ---- rule B
            const ConvGeom& g0 = c0->g;
            st.kind = ChainPlan::FUSED_B;
            // default: the direct VALU kernel (35 us per 1080p frame); SNNHIP_ESPCN_B=wino selects the Winograd / 4x4x1-MFMA kernel (45 us:
            // fewer instructions, but its 80 KB tile limits residency to 2 blocks per CU and the load phases of co-resident blocks coincide).
            // Variants measured in round 1 and removed (DESIGN.md section 5 keeps the findings): persistent + LDS-DMA double buffering 56-98 us,
            // two rows per thread 36 us (same as the default: neither LDS bandwidth nor the scalar weight loads were the limiter), persistent
            // with register prefetch 50 us, persistent Winograd with a quad-granular prefetch pipeline 50 us.
            const char* bmode = snnhip::option("SNNHIP_ESPCN_B");
            st.wino = bmode && strcmp(bmode, "wino") == 0;
            const int bTW = st.wino ? 64 : B_TW, bTH = st.wino ? 16 : B_TH;
            st.b = FusedBParams{g0.N, g0.H, g0.W, up_div(g0.W, bTW), up_div(g0.H, bTH), make_act_cfg(g0.act, g0.leaky), 0u, 0u};
            st.b.magicX = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(st.b.tilesX) - 1) / static_cast<unsigned>(st.b.tilesX));
            st.b.magicY = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(st.b.tilesY) - 1) / static_cast<unsigned>(st.b.tilesY));
            std::vector<float> wB(9 * 16 * 4);
            for (int tap = 0; tap < 9; ++tap)
                for (int ic = 0; ic < 16; ++ic)
                    for (int o = 0; o < 4; ++o) wB[(tap * 16 + ic) * 4 + o] = c0->w_oihw[(static_cast<size_t>(o) * 16 + ic) * 9 + tap];
            if (st.wino) {
                // U[pos][oc][ic] = (G g Gt)[xi][nu] as the kernel's LDS image: float index ((pos*4 + ic/4)*4 + oc)*4 + ic%4
                static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
                wB.assign(1024, 0.0f);
                for (int oc = 0; oc < 4; ++oc)
                    for (int ic = 0; ic < 16; ++ic) {
                        const float* gk = &c0->w_oihw[(static_cast<size_t>(oc) * 16 + ic) * 9];
                        double tmp[4][3];
                        for (int xi = 0; xi < 4; ++xi)
                            for (int v = 0; v < 3; ++v) tmp[xi][v] = Gm[xi][0] * gk[0 * 3 + v] + Gm[xi][1] * gk[1 * 3 + v] + Gm[xi][2] * gk[2 * 3 + v];
                        for (int xi = 0; xi < 4; ++xi)
                            for (int nu = 0; nu < 4; ++nu) {
                                const double u = tmp[xi][0] * Gm[nu][0] + tmp[xi][1] * Gm[nu][1] + tmp[xi][2] * Gm[nu][2];
                                wB[(((xi * 4 + nu) * 4 + (ic >> 2)) * 4 + oc) * 4 + (ic & 3)] = static_cast<float>(u);
                            }
                    }
            }
            std::vector<float> e1 = fold_epilogue(c0->epi4, 4, g0.useBN);
            rc = chain->upload(wB.data(), wB.size(), &st.w1);
            if (rc == SNNHIP_OK) rc = chain->upload(e1.data(), e1.size(), &st.e1);
            memcpy(st.outDims, sp1->outDims, sizeof(st.outDims));
            char buf[200];
            snprintf(buf, sizeof(buf), "fused[conv3x3(16->4)%s+depth_to_space(2)+tanh] %s tile=%dx%d kernel=%s", st.wino ? " winograd F(2x2,3x3)" : "",
                     st.wino ? "mfma_f32_4x4x1" : "valu_f32", bTW, bTH, st.wino ? "conv3x3_c16o4_wino_d2s_tanh_kernel" : "conv3x3_c16o4_d2s_tanh_kernel");
            st.desc = buf;
            st.flops = c0->flops;
            st.bytes = 4.0 * (static_cast<double>(g0.N) * g0.H * g0.W * (16 + 4) + 4.0 * 16 * 9);
            i += 2;
            ++fusedCount;
        } else if (snnhip_plan* spool = nullptr; i + 1 < n && c0 && !c0->depthwise && c0->g.kh == 7 && pool2d_plan_desc(plans[i + 1], nullptr) &&
                                          make_conv2d_stem32_pool_plan(ctx, plans[i], plans[i + 1], &spool) == SNNHIP_OK) {
            // ---- rule J: Conv2D 7x7 stride 2 (RGB) -> MaxPooling2D 3x3 stride 2 (the head of ResNet-18) -> the pooling runs in the stem's epilogue
            chain->owned.push_back(spool);
            st.kind = ChainPlan::PLAIN;
            st.plain = spool;
            memcpy(st.outDims, spool->outDims, sizeof(st.outDims));
            st.desc = spool->desc;
            st.flops = spool->flops;
            st.bytes = spool->bytes;
            i += 2;
            ++fusedCount;
        } else if (snnhip_plan* sirb = nullptr; i + 2 < n && c0 && c1 && c2 && !c0->depthwise && c0->g.kh == 3 && c0->g.IC == 3 && c1->depthwise && !c2->depthwise &&
                                         (make_stem_dwpw_march_plan(ctx, plans[i], plans[i + 1], plans[i + 2], &sirb) == SNNHIP_OK || // (large maps: the row-marching form)
                                          make_irb_plan(ctx, nullptr, plans[i + 1], plans[i + 2], nullptr, &sirb, plans[i]) == SNNHIP_OK)) {
            // ---- rule G with the network's stem as the 'expand' layer: Conv2D 3x3 (3 -> C channels) -> DepthwiseConv2D 3x3 -> Conv2D 1x1 (the head of
            // MobileNetV2) -> the same kernel, its staging gathers the 27 image values per pixel; the stem's output never reaches memory
            chain->owned.push_back(sirb);
            st.kind = ChainPlan::PLAIN;
            st.plain = sirb;
            memcpy(st.outDims, sirb->outDims, sizeof(st.outDims));
            st.desc = sirb->desc;
            st.flops = sirb->flops;
            st.bytes = sirb->bytes;
            i += 3;
            ++fusedCount;
        } else if (snnhip_plan* irb = nullptr; i + 2 < n && c0 && c1 && c2 && !c0->depthwise && c1->depthwise && !c2->depthwise &&
                                        make_irb_plan(ctx, plans[i], plans[i + 1], plans[i + 2], nullptr, &irb) == SNNHIP_OK) {
            // ---- rule G: Conv2D 1x1 -> DepthwiseConv2D 3x3 -> Conv2D 1x1 (an inverted-residual block without skip connection) -> one kernel
            chain->owned.push_back(irb);
            st.kind = ChainPlan::PLAIN;
            st.plain = irb;
            memcpy(st.outDims, irb->outDims, sizeof(st.outDims));
            st.desc = irb->desc;
            st.flops = irb->flops;
            st.bytes = irb->bytes;
            i += 3;
            ++fusedCount;
        } else if (snnhip_plan* dwpw = nullptr; i + 1 < n && c0 && c1 && c0->depthwise && !c1->depthwise &&
                                         (make_dwpw_march_plan(ctx, plans[i], plans[i + 1], &dwpw) == SNNHIP_OK || // (large stride-1 maps: the row-marching streaming form)
                                          make_irb_plan(ctx, nullptr, plans[i], plans[i + 1], nullptr, &dwpw) == SNNHIP_OK)) {
            // ---- rule G without an expand layer: DepthwiseConv2D 3x3 -> Conv2D 1x1 (MobileNetV2's first block) -> the same kernel, its hidden slice is the x tile
            chain->owned.push_back(dwpw);
            st.kind = ChainPlan::PLAIN;
            st.plain = dwpw;
            memcpy(st.outDims, dwpw->outDims, sizeof(st.outDims));
            st.desc = dwpw->desc;
            st.flops = dwpw->flops;
            st.bytes = dwpw->bytes;
            i += 2;
            ++fusedCount;
        } else if (auto* up = dynamic_cast<UpsamplePlanBase*>(plans[i]);
                   up && up->d.mode == SNNHIP_UPSAMPLE_NEAREST && up->d.scale == 2.0f && up->OH == 2 * up->d.H && up->OW == 2 * up->d.W && i + 1 < n &&
                   !snnhip::option("SNNHIP_NO_PAD_FUSION")) {
            // ---- rule D with a nearest x2 UpSampling2D in front: [UpSampling2D, Pad, Conv2D] or [UpSampling2D, Conv2D] -> one convolution launch
            auto* pd2 = dynamic_cast<PadPlanBase*>(plans[i + 1]);
            auto* cv = dynamic_cast<ConvPlanBase*>(plans[i + (pd2 ? 2 : 1) < n ? i + (pd2 ? 2 : 1) : i]);
            const int span = pd2 ? 3 : 2;
            snnhip_plan* fused = nullptr;
            if (cv && i + span <= n && !cv->depthwise && cv->g.preMode == 0 && cv->desc.rfind("conv2d_mfma", 0) == 0 && cv->g.N == up->d.N &&
                cv->g.IC == up->d.C && (pd2 ? (pd2->d.H == up->OH && pd2->d.W == up->OW && cv->g.H == pd2->OH && cv->g.W == pd2->OW)
                                            : (cv->g.H == up->OH && cv->g.W == up->OW))) {
                ConvGeom g2 = cv->g;
                g2.preMode = pd2 ? pd2->d.mode + 1 : SNNHIP_PAD_CONSTANT; // no Pad layer: an identity pad (offsets 0) in front of the upsampling
                g2.preX = pd2 ? pd2->d.padT : 0;
                g2.preY = pd2 ? pd2->d.padL : 0;
                g2.preShift = 1;
                g2.srcH = up->d.H;
                g2.srcW = up->d.W;
                if (make_conv2d_mfma_plan(ctx, g2, cv->w_oihw.data(), cv->epi4, &fused) != SNNHIP_OK) fused = nullptr;
            }
            st.kind = ChainPlan::PLAIN;
            if (fused) {
                chain->owned.push_back(fused);
                st.plain = fused;
                i += span;
                ++fusedCount;
            } else {
                st.plain = plans[i];
                i += 1;
            }
            memcpy(st.outDims, st.plain->outDims, sizeof(st.outDims));
            st.desc = st.plain->desc;
            st.flops = st.plain->flops;
            st.bytes = st.plain->bytes;
        } else if (auto* pd = dynamic_cast<PadPlanBase*>(plans[i]); pd && c1 && !c1->depthwise && c1->g.preMode == 0 && c1->g.N == pd->d.N &&
                   c1->g.H == pd->OH && c1->g.W == pd->OW && c1->g.IC == pd->d.C &&
                   (c1->desc.rfind("conv2d_mfma", 0) == 0 || c1->desc.rfind("conv2d_rowfold", 0) == 0) && !snnhip::option("SNNHIP_NO_PAD_FUSION")) {
            // ---- rule D: Pad + Conv2D -> the convolution stages its tiles straight from the unpadded tensor ("reflect Pad,
            // better fused into the following conv's load stage"); only the MFMA kernel has the pre-pad address path, so a convolution that was
            // routed to another kernel (the channel-thin image-producing layers) keeps its separate Pad launch
            ConvGeom g2 = c1->g;
            g2.preMode = pd->d.mode + 1; // pad desc 0/1/2 = constant / replicate / reflect -> SNNHIP_PAD_CONSTANT / _REPLICATE / _REFLECT
            g2.preX = pd->d.padT;        // sic: the Pad shader shifts x by the top pad and y by the left pad (padlayerVulkan.cpp:81-82)
            g2.preY = pd->d.padL;
            g2.srcH = pd->d.H;
            g2.srcW = pd->d.W;
            snnhip_plan* fused = nullptr;
            const int frc = c1->desc.rfind("conv2d_rowfold", 0) == 0 ? make_conv2d_rowfold_plan(ctx, g2, c1->w_oihw.data(), c1->epi4, &fused)
                                                                      : make_conv2d_mfma_plan(ctx, g2, c1->w_oihw.data(), c1->epi4, &fused);
            if (frc == SNNHIP_OK) {
                fused->ctx = ctx;
                chain->owned.push_back(fused);
                st.kind = ChainPlan::PLAIN;
                st.plain = fused;
                memcpy(st.outDims, fused->outDims, sizeof(st.outDims));
                st.desc = fused->desc;
                st.flops = fused->flops;
                st.bytes = fused->bytes;
                i += 2;
                ++fusedCount;
            } else {
                st.kind = ChainPlan::PLAIN;
                st.plain = plans[i];
                memcpy(st.outDims, plans[i]->outDims, sizeof(st.outDims));
                st.desc = plans[i]->desc;
                st.flops = plans[i]->flops;
                st.bytes = plans[i]->bytes;
                i += 1;
            }
        } else {
            st.kind = ChainPlan::PLAIN;
            st.plain = plans[i];
            memcpy(st.outDims, plans[i]->outDims, sizeof(st.outDims));
            st.desc = plans[i]->desc;
            st.flops = plans[i]->flops;
            st.bytes = plans[i]->bytes;
            i += 1;
        }
        chain->steps.push_back(st);
    }
    // ---- rule I: an InstanceNorm step followed by a convolution step (as given, or built by rule D above) whose kernel can normalise in its
    // staging (today: conv2d_mfma's fp16 kernels).  SNNHIP_NO_NORM_FOLD keeps the norm's own normalise sweep.
    for (size_t k = 0; rc == SNNHIP_OK && k + 1 < chain->steps.size() && !snnhip::option("SNNHIP_NO_NORM_FOLD"); ++k) {
        ChainPlan::Step &a = chain->steps[k], &b = chain->steps[k + 1];
        if (a.kind != ChainPlan::PLAIN || b.kind != ChainPlan::PLAIN) continue;
        snnhip_instancenorm_desc nd;
        auto* cv = dynamic_cast<ConvPlanBase*>(b.plain);
        if (!cv || cv->depthwise || cv->numInputs != 1 || cv->g.normShift || !instancenorm_plan_desc(a.plain, &nd) || !act_is_simple(nd.act)) continue;
        if (nd.N != cv->inDims[0] || nd.H != cv->inDims[1] || nd.W != cv->inDims[2] || nd.C != cv->inDims[3]) continue;
        // only where the convolution already runs on a kernel that can normalise (trading conv2d_wide_f16 for the 128-pixel kernel cost more than
        // the normalise sweep saves, measured on Candy's residual blocks: 160 + 290 us apart, 600 us folded -- the wide kernel has its own form now)
        // conv2d_wide_f16 normalises in LDS behind its DMA: worth it on the 64 / 128-channel blocks (body layers 310 + 130 us apart -> 370 us), not
        // behind a fused UpSampling (the pass runs on the 4x replicated pixels) nor on the VALU-bound 32-channel blocks (64 -> 32 up-conv: 1.07 ms
        // + 0.17 ms sweep apart, 1.70 ms folded)
        const bool onWide = cv->desc.rfind("conv2d_mfma_wide_f16", 0) == 0 && !cv->g.preShift && cv->g.OC % 64 == 0;
        // conv2d_upconv (round 6) stages the LOW-RESOLUTION tensor, so its pass runs once per pixel: the 64 -> 32 up-convolution of the style graphs reads the
        // norm's input and the 944 MB normalise sweep in front of it disappears
        const bool onUpconv = cv->desc.rfind("conv2d_mfma_upconv_f16", 0) == 0 && cv->g.IC == 64;
        if (cv->desc.rfind("conv2d_mfma_f16_", 0) != 0 && cv->desc.rfind("conv2d_rowfold", 0) != 0 && !(onWide && !snnhip::option("SNNHIP_NO_WIDE_NORM")) && !onUpconv) continue;
        ConvGeom g2 = cv->g;
        if (!instancenorm_stat_pointers(a.plain, &g2.normShift, &g2.normMul)) continue;
        g2.normAct = nd.act;
        g2.normLeaky = nd.leaky;
        snnhip_plan* fused = nullptr;
        const int frc = cv->desc.rfind("conv2d_rowfold", 0) == 0 ? make_conv2d_rowfold_plan(ctx, g2, cv->w_oihw.data(), cv->epi4, &fused)
                                                                 : make_conv2d_mfma_plan(ctx, g2, cv->w_oihw.data(), cv->epi4, &fused);
        if (frc != SNNHIP_OK) continue;
        if ((onWide && fused->desc.find("conv2d_mfma_wide_f16") == std::string::npos) || (onUpconv && fused->desc.find("conv2d_mfma_upconv_f16") == std::string::npos)) { // (routed elsewhere with the norm attached: keep the separate launches)
            delete fused;
            continue;
        }
        chain->owned.push_back(fused);
        auto* both = new InstanceNormConvPlan();
        both->ctx = ctx;
        both->norm = a.plain;
        both->conv = fused;
        both->dtype = fused->dtype;
        both->numInputs = fused->numInputs;
        memcpy(both->inDims, fused->inDims, sizeof(both->inDims));
        memcpy(both->outDims, fused->outDims, sizeof(both->outDims));
        both->flops = a.flops + b.flops;
        both->bytes = a.bytes / 3.0 + b.bytes; // the norm's normalise sweep (one read + one write of its three passes) is gone
        both->desc = "instancenorm(statistics sweep + fold) -> " + fused->desc;
        chain->owned.push_back(both);
        a.plain = both;
        a.desc = both->desc;
        a.flops = both->flops;
        a.bytes = both->bytes;
        memcpy(a.outDims, b.outDims, sizeof(a.outDims));
        chain->steps.erase(chain->steps.begin() + static_cast<long>(k) + 1);
        ++fusedCount;
    }
    // ---- rule F: a convolution step (as given, or one a rule above built) whose kernel can reduce its output tiles to {mean, M2} records, followed by
    // a step that starts with an InstanceNorm: the norm's statistics sweep becomes a fold over those records.  The consumer is the norm itself (the
    // two steps become one: conv, fold, normalise in place), the norm + Add of rule H (last step of a two-input chain), or the norm -> convolution of
    // rule I.  Default: conv2d_wide_f16, conv2d_upconv and conv2d_s2march only -- their 256 / 512-pixel tiles pass through registers on their way out anyway (+3 % on the kernel, one
    // tensor read saved).  SNNHIP_NORM_FUSION=1 also takes conv2d_mfma's fp16 kernel (measured a loss: its short blocks pay 45-125 us per layer
    // for the statistics where the sweep costs 40), =0 switches the rule off.
    const char* normFusion = snnhip::option("SNNHIP_NORM_FUSION");
    const int normFusionMode = normFusion ? atoi(normFusion) : -1; // -1 default, 0 off, 1 every kernel that can
    for (size_t k = 0; rc == SNNHIP_OK && k + 1 < chain->steps.size() && normFusionMode != 0; ++k) {
        ChainPlan::Step &a = chain->steps[k], &b = chain->steps[k + 1];
        if (a.kind != ChainPlan::PLAIN || b.kind != ChainPlan::PLAIN) continue;
        auto* aIn = dynamic_cast<InstanceNormConvPlan*>(a.plain); // the producer may itself be a normalising convolution (rule I): its inner plan is the chain's
        auto* cv = dynamic_cast<ConvPlanBase*>(aIn ? aIn->conv : a.plain);
        if (!cv || cv->depthwise || cv->numInputs != 1) continue;
        if (normFusionMode < 0 && cv->desc.find("conv2d_mfma_wide_f16") == std::string::npos && cv->desc.find("conv2d_mfma_upconv_f16") == std::string::npos &&
            !(cv->desc.find("conv2d_mfma_stem_f16") != std::string::npos && !snnhip::option("SNNHIP_STEM_NO_STATS")) &&
            !(cv->desc.find("row-marching") != std::string::npos && cv->desc.find(" s=2 ") != std::string::npos))
            continue;
        {
            // small tensors (one 720p image: 17 MB per layer) are swept out of the L2 / MALL in less time than the two fold launches take
            // (Candy batch 1: 1.12 ms without the rule, 1.24 ms with it); from a few images per batch on the sweep is an HBM pass
            const char* mb = snnhip::option("SNNHIP_NORM_FUSION_MIN_MB");
            const double minBytes = (mb ? atof(mb) : 128.0) * 1048576.0;
            const double outBytes = static_cast<double>(cv->outDims[0]) * cv->outDims[1] * cv->outDims[2] * cv->outDims[3] * (cv->dtype == SNNHIP_F16 ? 2.0 : 4.0);
            if (normFusionMode < 0 && outBytes < minBytes) continue;
        }
        auto* inConv = dynamic_cast<InstanceNormConvPlan*>(b.plain);
        snnhip_plan* normPlan = inConv ? inConv->norm : b.plain;
        snnhip_instancenorm_desc nd;
        bool normAdd = false;
        if (!instancenorm_plan_desc(normPlan, &nd)) {
            normPlan = instancenorm_add_use_tile_stats(b.plain, TileStatsRef()); // probe: which norm (the reference stays empty = a sweep)
            if (!normPlan || !instancenorm_plan_desc(normPlan, &nd)) continue;
            normAdd = true;
        }
        if (nd.N != cv->outDims[0] || nd.H != cv->outDims[1] || nd.W != cv->outDims[2] || nd.C != cv->outDims[3]) continue;
        // The statistics epilogue changes the convolution plan (tile-stat buffer, LDS size, description): never switch it on in a plan the
        // caller owns -- a borrowed per-layer plan stays what it was; the chain works on its own copy (rule D's product already is chain-owned).
        bool borrowed = false;
        for (int i = 0; i < n; ++i) borrowed = borrowed || plans[i] == a.plain;
        if (borrowed && !aIn) {
            snnhip_plan* copy = nullptr;
            if (make_conv2d_mfma_plan(ctx, cv->g, cv->w_oihw.data(), cv->epi4, &copy) != SNNHIP_OK) continue;
            auto* cc = dynamic_cast<ConvPlanBase*>(copy);
            if (!cc || !cc->enableTileStats()) {
                delete copy;
                continue;
            }
            chain->owned.push_back(copy);
            cv = cc;
        } else if (!cv->enableTileStats()) {
            continue;
        }
        // the fold scratch of the norm is sized here, at plan creation: an allocation inside run() would break a hipGraph capture in progress
        if (instancenorm_reserve_tile_stats(normPlan, cv->statTilesX, cv->statTilesY) != SNNHIP_OK) continue;
        TileStatsRef tiles;
        tiles.part = cv->statPart; tiles.tilesX = cv->statTilesX; tiles.tilesY = cv->statTilesY; tiles.TH = cv->statTH; tiles.TW = cv->statTW;
        // a kernel that folds the records itself (the last block of an image: norm_fold.h) leaves nothing to launch between it and the consumer
        NormFoldTarget target;
        if (!snnhip::option("SNNHIP_NO_KERNEL_FOLD") && instancenorm_fold_target(normPlan, &target)) tiles.folded = cv->enableNormFold(target);
        if (!tiles.folded && cv->tileStatsNeedKernelFold()) { // (an allocation failed): per-block records have no fold launch -- the norm keeps its sweep
            cv->disableTileStats();
            continue;
        }
        if (aIn) {
            const size_t arrow = aIn->desc.find(" -> ");
            aIn->desc = (arrow == std::string::npos ? std::string("instancenorm") : aIn->desc.substr(0, arrow)) + " -> " + cv->desc;
            a.desc = aIn->desc;
        } else {
            a.plain = cv;
            a.desc = cv->desc;
        }
        ++fusedCount;
        if (normAdd) { // rules F + H: the plan was built by the graph walk for this chain (it is the chain's to change)
            instancenorm_add_use_tile_stats(b.plain, tiles);
            b.desc = b.plain->desc;
            b.bytes *= 0.75; // the statistics sweep (one read of its four passes) is gone
            continue;
        }
        if (inConv) { // rules F + I
            inConv->tiles = tiles;
            inConv->desc = (tiles.folded ? "instancenorm(statistics from the convolution in front) -> " : "instancenorm(fold of tile stats) -> ") + inConv->conv->desc;
            b.desc = inConv->desc;
            b.bytes -= static_cast<double>(nd.N) * nd.H * nd.W * nd.C * (cv->dtype == SNNHIP_F16 ? 2.0 : 4.0);
            continue;
        }
        auto* both = new ConvInstanceNormPlan();
        both->ctx = ctx;
        both->conv = aIn ? static_cast<snnhip_plan*>(aIn) : cv;
        both->norm = b.plain;
        both->tiles = tiles;
        both->dtype = cv->dtype;
        memcpy(both->inDims, cv->inDims, sizeof(both->inDims));
        memcpy(both->outDims, cv->outDims, sizeof(both->outDims));
        both->flops = a.flops + b.flops;
        both->bytes = a.bytes + b.bytes * 2.0 / 3.0;
        both->desc = a.desc + (tiles.folded ? " -> instancenorm(1 sweep) act=" : " -> instancenorm(fold of tile stats + 1 sweep) act=") + std::to_string(nd.act);
        chain->owned.push_back(both);
        a.plain = both;
        a.desc = both->desc;
        a.flops = both->flops;
        a.bytes = both->bytes;
        chain->steps.erase(chain->steps.begin() + static_cast<long>(k) + 1);
    }
    if (rc == SNNHIP_OK && fusedCount == 0) {
        set_error("chain fusion: no rule matches these %d plans", n);
        rc = SNNHIP_E_UNSUPPORTED;
    }
    // element type of the chain = that of its convolutions (the element-wise plans adapt to the tensors they are given)
    for (int i = 0; i < n; ++i)
        if (auto* c = dynamic_cast<ConvPlanBase*>(plans[i])) {
            chain->dtype = c->g.dtype;
            break;
        }
    for (size_t i = 0; rc == SNNHIP_OK && i + 1 < chain->steps.size(); ++i) {
        snnhip_tensor* t = nullptr;
        const int* d = chain->steps[i].outDims;
        rc = snnhip_tensor_alloc(ctx, d[0], d[1], d[2], d[3], chain->dtype, &t);
        if (rc == SNNHIP_OK) chain->mids.push_back(t);
    }
    if (rc != SNNHIP_OK) {
        delete chain;
        return rc;
    }
    for (int i = 0; i < n; ++i) {
        chain->flops += plans[i]->flops;
        chain->bytes += plans[i]->bytes;
    }
    std::string d = "chain{";
    for (size_t i = 0; i < chain->steps.size(); ++i) d += (i ? " -> " : "") + chain->steps[i].desc;
    chain->desc = d + "}";
    *out = chain;
    return SNNHIP_OK;
}

// graph walk: a plan it built for this chain (the InstanceNorm -> Add of rule H at the chain's end) becomes the chain's to delete
bool chain_adopt_plan(snnhip_plan* chain, snnhip_plan* p) {
    auto* c = dynamic_cast<ChainPlan*>(chain);
    if (!c) return false;
    c->owned.push_back(p);
    return true;
}

} // namespace snnhip
