// conv2d_stem_f32.hip -- fp32 k x k convolution of a channel-thin image (IC <= 4) with 32 | OC: the RGB stems (ResNet-18 7x7 stride 2, MobileNetV2 /
// YOLO 3x3 stride 2, BASELINE configs[0]'s 3x3 stride 1) on v_mfma_f32_32x32x2_f32.
//
// conv2d_mfma_kernel's tap-pair mode walks the taps with a rolled loop and one LDS operand read per MFMA: the 7x7 stem of ResNet-18 (batch 32) ran
// at 58 TF/s, 130 us, bound by instruction issue.  The structure of conv2d_stem_f16.hip, for fp32:
//   * lane half h reads the 8 bytes {c_h, c_(2+h)} of a tap's pixel with ONE ds_read_b64 and feeds two MFMAs with them (K = 2 per MFMA: channel
//     pair (0,1), then (2,3)); 2 MFMAs per tap, 3 of 4 operand slots carry data.  The tile sits in LDS as PLANES [row][h][column parity][column / 2]
//     of those 8-byte pairs (stride 1: no parity split): the 32 lanes of an operand read walk 32 CONSECUTIVE pairs = all 64 banks once.  (Round 2
//     stored a pixel as 16 bytes {c0, c2, c1, c3}: at stride 2 the lanes sat 32 bytes apart, 8 bank pairs for 32 lanes -- PMC: SQ_LDS_BANK_CONFLICT
//     65 % of SQ_LDS_IDX_ACTIVE on the ResNet-18 stem, 51 % on MobileNetV2's.)
//   * ALL weights of a 32-channel output block in registers (2 k^2 floats per lane: the MFMA's A operand is one VGPR), loaded once per wave;
//   * a wave owns NR = 4 output rows x 32 columns: the operand of (input row r, tap column fx) is read ONCE and feeds the MFMAs of every output
//     row y with r - s y a valid kernel row; straight-line code;
//   * epilogue per output row through a wave-private LDS scratch: the lane's 16-byte channel runs in, 64 contiguous bytes per lane out (whole lines).
// Same operator contract as the other convolution kernels (padding modes, bias -> BN -> activation).
#include "epilogue.h"
#include "snnhip_internal.h"

#include <cstring>

namespace snnhip {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Stem32Params {
    int N, H, W, IC, OC, padx, pady, padMode, useBN, OH, OW;
    int tilesX, tilesY;
    int poolOH, poolOW; // dense kernel, POOL form: dims of the MaxPooling2D 3x3 stride 2 output written instead of the convolution's (0 otherwise)
};

constexpr int kTW = 32;                 // block tile: 4 waves stacked in y, NR output rows per wave (4; 2 on grids that leave CUs idle: half the latency)
// 8-byte pairs per plane row: the 32 pairs of an operand read + the taps' reach, as tight as the kernel allows (the 3x3 stride-2 tile must leave room for
// three blocks per CU: 40 pairs per row cost MobileNetV2's stem 161 -> 175 us)
__host__ __device__ constexpr int plane_w(int K, int S) { return S == 2 ? (K == 7 ? 36 : 33) : 32 + K - 1; }
constexpr int kDensePW = 36;            // dense 7x7 stride-2 kernel: floats per column-parity row of a channel plane (35 even columns of the 69-column halo tile)
constexpr int kDenseSteps = 74;         // ... K steps (= weight registers) per output row: ceil(7 * 7 * 3 / 2)
constexpr int kOutPitch = 36;           // floats per pixel row of a wave's output scratch (32 + 4: 16-byte aligned, the runs of 8 lanes on distinct banks)


// epilogue of a wave's kNR output rows x 32 columns: acc[yy][4g + k] = channel 8g + 4h + k of pixel (row yy, column l32); e = the lane's 16 rows of the
// epilogue table; sc = the wave's LDS scratch
template <int kNR, bool SIMPLE>
__device__ __forceinline__ void stem32_epilogue(const Stem32Params& p, const ActCfg& ac, const f32x16 (&acc)[kNR], const float4 (&e)[16], float* const sc, float* __restrict__ y, int n,
                                                int oyW, int ox0, int lane) {
    const int l32 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int yy = 0; yy < kNR; ++yy) {
        const int oy = oyW + yy;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = epi_affine(acc[yy][4 * g + k], e[4 * g + k], p.useBN);
                o[k] = SIMPLE ? apply_act<true>(ac, v, 0.0f) : epi_act(ac.act, ac.leaky, v, 0.0f);
            }
            *reinterpret_cast<float4*>(sc + l32 * kOutPitch + 8 * g + 4 * h) = make_float4(o[0], o[1], o[2], o[3]);
        }
        // (wave-private scratch: a wave's LDS operations complete in order).  Out: 16-byte pieces, lane L = piece L % 8 of pixel L / 8 (+ 8 i)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int opix = 8 * i + (lane >> 3), piece = lane & 7;
            const float4 v = *reinterpret_cast<const float4*>(sc + opix * kOutPitch + 4 * piece);
            const int ox = ox0 + opix;
            if (oy < p.OH && ox < p.OW) *reinterpret_cast<float4*>(y + (static_cast<size_t>(n * p.OH + oy) * p.OW + ox) * p.OC + blockIdx.y * 32 + 4 * piece) = v;
        }
    }
}

template <int K, int S, int kNR, bool SIMPLE>
__global__ __launch_bounds__(256, K <= 3 ? 3 : 2) void conv2d_stem32_kernel(Stem32Params p, ActCfg ac, const float* __restrict__ x, const float* __restrict__ wp,
                                                            const float4* __restrict__ epi, float* __restrict__ y) {
    constexpr int kTH = 4 * kNR;
    constexpr int IN_H = (kTH - 1) * S + K, IN_W = (kTW - 1) * S + K; // staged halo tile
    constexpr int ROWS = (kNR - 1) * S + K;                           // input rows a wave touches
    constexpr int NW = 2 * K * K;                                     // weight registers per lane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NP = S == 2 ? 2 : 1;                                // column-parity planes
    constexpr int kPlaneW = plane_w(K, S);
    constexpr int ROWFL = 2 * NP * kPlaneW * 2;                       // floats per tile row: [h][parity][kPlaneW] pairs
    float* const tile = smem;                                         // [IN_H][2][NP][kPlaneW][2]
    float* const oscr = smem + IN_H * ROWFL;                          // [4 waves][32][kOutPitch]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, h = lane >> 5;
    const int mt = blockIdx.x;
    const int tx = mt % p.tilesX, ty = (mt / p.tilesX) % p.tilesY, n = mt / (p.tilesX * p.tilesY);
    const int ox0 = tx * kTW, oy0 = ty * kTH;

    // ---- weights: step s = 2 (fy K + fx) + m: lane (oc = l32, half h) holds W[oc][ic = 2m + h][fy][fx] (0 for ic >= IC)
    float wa[NW];
    {
        const float* wsrc = wp + static_cast<size_t>(blockIdx.y) * NW * 64 + lane;
#pragma unroll
        for (int s = 0; s < NW; ++s) wa[s] = wsrc[s * 64];
    }

    // the lane's 16 rows of the epilogue table, requested up front (at the top of the epilogue they were 16 loads the wave sat out: an L2 round trip
    // in a block that lives for a few microseconds)
    float4 e[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) e[4 * g + k] = epi[blockIdx.y * 32 + 8 * g + 4 * h + k];

    // ---- stage the halo tile into the pair planes; every load of the thread is issued before its first LDS write
    constexpr int kR = (IN_H * IN_W + 255) / 256;
    {
        float sv[kR][4];
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int e = tid + 256 * r;
            const int rr = e / IN_W, c = e - rr * IN_W;
            const int sy = resolve_nobranch(oy0 * S - p.pady + rr, p.H, p.padMode);
            const int sx = resolve_nobranch(ox0 * S - p.padx + c, p.W, p.padMode);
            const bool ok = e < IN_H * IN_W && sy >= 0 && sx >= 0;
            const float* src = x + (static_cast<size_t>(n * p.H + (ok ? sy : 0)) * p.W + (ok ? sx : 0)) * p.IC;
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[r][k] = (ok && k < p.IC) ? src[k] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int e = tid + 256 * r;
            const int rr = e / IN_W, c = e - rr * IN_W;
            if (e < IN_H * IN_W) {
                float* const d0 = tile + rr * ROWFL + ((S == 2 ? (c & 1) * kPlaneW + (c >> 1) : c)) * 2; // plane h = 0: {c0, c2}
                *reinterpret_cast<float2*>(d0) = make_float2(sv[r][0], sv[r][2]);
                *reinterpret_cast<float2*>(d0 + NP * kPlaneW * 2) = make_float2(sv[r][1], sv[r][3]); // plane h = 1: {c1, c3}
            }
        }
    }
    __syncthreads();

    f32x16 acc[kNR];
#pragma unroll
    for (int i = 0; i < kNR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // operand base of this lane: row wave*kNR*S, plane h, pair l32 (tap column fx adds parity plane fx & 1 and fx / 2 pairs: immediates)
    const float* const tb = tile + (wave * kNR * S) * ROWFL + (h * NP * kPlaneW + l32) * 2;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int fx = 0; fx < K; ++fx) {
            const float2 b = *reinterpret_cast<const float2*>(tb + r * ROWFL + (S == 2 ? (fx & 1) * kPlaneW + (fx >> 1) : fx) * 2);
#pragma unroll
            for (int yy = 0; yy < kNR; ++yy) {
                const int fy = r - yy * S;
                if (fy < 0 || fy >= K) continue; // compile-time after unrolling
                acc[yy] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[2 * (fy * K + fx)], b.x, acc[yy], 0, 0, 0);
                acc[yy] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[2 * (fy * K + fx) + 1], b.y, acc[yy], 0, 0, 0);
            }
        }

    stem32_epilogue<kNR, SIMPLE>(p, ac, acc, e, oscr + wave * (32 * kOutPitch), y, n, oy0 + wave * kNR, ox0, lane);
}

// The 7x7 stride-2 RGB stem (ResNet-18) with the reduction packed DENSELY: K = 7 x 7 taps x 3 channels = 147 -> 74 K steps of 2 per output row instead of the 98
// of the kernel above (which spends a K slot on the absent fourth channel of every tap); the stem is bound by the matrix pipe, so the MFMA count is its time.
//   * K order: e = (fy 7 + fx) 3 + ch; the lane halves of an MFMA take e = 2k and 2k + 1.  Seen from the input, the elements of the wave's ROWS input rows
//     are g = r 21 + fx 3 + ch and an output row yy reads them at e = g - 42 yy: an EVEN shift, so the pairing (2s, 2s + 1) of the input elements is the
//     same for every output row -- the operand of step s is read ONCE and feeds the MFMAs of up to four output rows, with weight register s - 21 yy.
//   * tile in LDS as channel planes [ch][row][column parity][column / 2] of floats: the 32 lanes of a half read 32 consecutive floats; the two halves read
//     different (tap, channel) elements, i.e. different compile-time offsets: one v_cndmask + one ds_read_b32 per step.
//   * 74 weight registers per lane (flat order, zero behind e = 146); epilogue = the kernel above.
//   * POOL (chain rule J): the MaxPooling2D 3x3 stride 2 behind the stem (no padding on top / left: maxpool2dVulkan.cpp:62-64; window clipped at the
//     bottom / right, the maximum starts at -100000.0) runs in the epilogue and the 112x112 tensor never reaches memory.  A block then owns 7 x 15
//     POOLED pixels = rows [14 ty, 14 ty + 15) x columns [30 tx, 30 tx + 31) of the convolution (one row / column of its 16 x 32 tile is spare, 8
//     instead of 7 row tiles: +14 % stem work against the pool launch's 29 us and 183 MB).  Wave w holds rows 4w .. 4w + 3: pooled row 2w is the
//     maximum of its rows 0-2, pooled row 2w + 1 needs row 0 of wave w + 1, handed over through the (by then dead) input tile; the column maximum is
//     taken while a pooled row passes through the wave's output scratch.
template <int kNR, bool SIMPLE, bool POOL = false>
__global__ __launch_bounds__(256, 3) void conv2d_stem32_dense_kernel(Stem32Params p, ActCfg ac, const float* __restrict__ x, const float* __restrict__ wp,
                                                                  const float4* __restrict__ epi, float* __restrict__ y) {
    constexpr int K = 7, S = 2;
    constexpr int kTH = 4 * kNR;
    constexpr int IN_H = (kTH - 1) * S + K, IN_W = (kTW - 1) * S + K; // staged halo tile
    constexpr int ROWS = (kNR - 1) * S + K;                           // input rows a wave touches
    constexpr int PW = kDensePW, ROWFL = 2 * PW, CHFL = IN_H * ROWFL; // floats per parity row / per (channel, row) / per channel plane
    constexpr int NE = ROWS * 21, NS = (NE + 1) / 2;                  // input elements (row, fx, ch) of a wave, K steps
    constexpr int NWD = kDenseSteps;                                  // weight registers: pairs of the flat 147 (+1 zero)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const tile = smem;                 // [3][IN_H][2][PW]
    float* const oscr = smem + 3 * CHFL;      // [4 waves][32][kOutPitch]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, h = lane >> 5;
    const int mt = blockIdx.x;
    const int tx = mt % p.tilesX, ty = (mt / p.tilesX) % p.tilesY, n = mt / (p.tilesX * p.tilesY);
    const int ox0 = POOL ? tx * 30 : tx * kTW, oy0 = POOL ? ty * 14 : ty * kTH;
    static_assert(!POOL || kNR == 4, "the pooling epilogue is written for 16-row tiles");

    float wa[NWD];
    {
        const float* wsrc = wp + static_cast<size_t>(blockIdx.y) * NWD * 64 + lane;
#pragma unroll
        for (int s = 0; s < NWD; ++s) wa[s] = wsrc[s * 64];
    }
    // the block's 32 rows of the epilogue table go through LDS (requested here, read back after the MFMA loop): held in registers across the loop -- 64 of
    // them per lane, as in the kernel above -- they were the difference between two and three resident blocks per CU
    float4* const etab = reinterpret_cast<float4*>(smem + 3 * CHFL + 4 * 32 * kOutPitch); // [32]
    if (tid < 32) etab[tid] = epi[blockIdx.y * 32 + tid];

    // ---- stage the halo tile into the channel planes; every load of the thread is issued before its first LDS write
    constexpr int kR = (IN_H * IN_W + 255) / 256;
    {
        float sv[kR][3];
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int el = tid + 256 * r;
            const int rr = el / IN_W, c = el - rr * IN_W;
            const int sy = resolve_nobranch(oy0 * S - p.pady + rr, p.H, p.padMode);
            const int sx = resolve_nobranch(ox0 * S - p.padx + c, p.W, p.padMode);
            const bool ok = el < IN_H * IN_W && sy >= 0 && sx >= 0;
            const float* src = x + (static_cast<size_t>(n * p.H + (ok ? sy : 0)) * p.W + (ok ? sx : 0)) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float t = src[k];
                sv[r][k] = ok ? t : 0.0f;
            }
        }
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int el = tid + 256 * r;
            const int rr = el / IN_W, c = el - rr * IN_W;
            if (el < IN_H * IN_W) {
                float* const d0 = tile + rr * ROWFL + (c & 1) * PW + (c >> 1);
#pragma unroll
                for (int k = 0; k < 3; ++k) d0[k * CHFL] = sv[r][k];
            }
        }
    }
    __syncthreads();

    f32x16 acc[kNR];
#pragma unroll
    for (int i = 0; i < kNR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const float* const tb = tile + (wave * kNR * S) * ROWFL + l32;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        // input elements 2s (lane half 0) and 2s + 1 (half 1; the one past the end repeats the last: its weight is zero): (row, fx, ch) -> plane offset
        const int g0 = 2 * s, g1 = 2 * s + 1 < NE ? 2 * s + 1 : NE - 1;
        const int r0 = g0 / 21, fx0 = (g0 % 21) / 3, c0 = g0 % 3;
        const int r1 = g1 / 21, fx1 = (g1 % 21) / 3, c1 = g1 % 3;
        const int o0 = c0 * CHFL + r0 * ROWFL + (fx0 & 1) * PW + (fx0 >> 1);
        const int o1 = c1 * CHFL + r1 * ROWFL + (fx1 & 1) * PW + (fx1 >> 1);
        const float b = tb[h ? o1 : o0];
#pragma unroll
        for (int yy = 0; yy < kNR; ++yy) {
            const int k = s - 21 * yy; // weight pair of output row yy
            if (k < 0 || k >= NWD) continue; // compile-time after unrolling
            acc[yy] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[k], b, acc[yy], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0); // (nothing of the epilogue is scheduled into the K loop: its registers are spoken for)
    if constexpr (!POOL) {
        float4 e[16];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) e[4 * g + kk] = etab[8 * g + 4 * h + kk];
        stem32_epilogue<kNR, SIMPLE>(p, ac, acc, e, oscr + wave * (32 * kOutPitch), y, n, oy0 + wave * kNR, ox0, lane);
    } else {
        constexpr float kPoolFloor = -100000.0f; // the reference's initial maximum; also the value of a pixel outside the convolution's output
        const bool colOk = ox0 + l32 < p.OW;
#pragma unroll
        for (int g = 0; g < 4; ++g) { // four table rows at a time (all sixteen next to the 64 accumulators and the hand-over row spilled)
            float4 e[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) e[kk] = etab[8 * g + 4 * h + kk];
#pragma unroll
            for (int yy = 0; yy < kNR; ++yy) {
                const bool ok = colOk && oy0 + wave * kNR + yy < p.OH;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const float v = epi_affine(acc[yy][4 * g + kk], e[kk], p.useBN);
                    const float a = SIMPLE ? apply_act<true>(ac, v, 0.0f) : epi_act(ac.act, ac.leaky, v, 0.0f);
                    acc[yy][4 * g + kk] = ok ? a : kPoolFloor;
                }
            }
        }
        __syncthreads(); // every wave is done with the input tile: its space carries the hand-over rows
        float* const xrow = tile + wave * (32 * kOutPitch);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(xrow + l32 * kOutPitch + 8 * g + 4 * h) = make_float4(acc[0][4 * g], acc[0][4 * g + 1], acc[0][4 * g + 2], acc[0][4 * g + 3]);
        __syncthreads();
        const float* const nrow = tile + ((wave + 1) & 3) * (32 * kOutPitch) + l32 * kOutPitch + 4 * h; // row 0 of the wave below (wave 3: unused)
        float* const sc = oscr + wave * (32 * kOutPitch);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (jj == 1 && wave == 3) break; // pooled row 7 of the tile belongs to the next one (wave-uniform)
            const int py = ty * 7 + 2 * wave + jj;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float m[4];
                const float4 nx4 = jj == 1 ? *reinterpret_cast<const float4*>(nrow + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float nx[4] = {nx4.x, nx4.y, nx4.z, nx4.w};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int i = 4 * g + kk;
                    const float t = jj == 0 ? fmaxf(fmaxf(acc[0][i], acc[1][i]), acc[2][i]) : fmaxf(fmaxf(acc[2][i], acc[3][i]), nx[kk]);
                    m[kk] = fmaxf(t, kPoolFloor);
                }
                *reinterpret_cast<float4*>(sc + l32 * kOutPitch + 8 * g + 4 * h) = make_float4(m[0], m[1], m[2], m[3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (wave-private scratch: LDS operations of a wave complete in order; keep the compiler's order)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) { // 15 pooled pixels x 8 16-byte pieces = 120 items over two passes of the wave
                const int item = lane + 64 * pass, opix = item >> 3, piece = item & 7;
                if (item < 120) {
                    const float4 a = *reinterpret_cast<const float4*>(sc + (2 * opix) * kOutPitch + 4 * piece);
                    const float4 b = *reinterpret_cast<const float4*>(sc + (2 * opix + 1) * kOutPitch + 4 * piece);
                    const float4 c = *reinterpret_cast<const float4*>(sc + (2 * opix + 2) * kOutPitch + 4 * piece);
                    const int px = tx * 15 + opix;
                    if (py < p.poolOH && px < p.poolOW)
                        *reinterpret_cast<float4*>(y + (static_cast<size_t>(n * p.poolOH + py) * p.poolOW + px) * p.OC + blockIdx.y * 32 + 4 * piece) =
                            make_float4(fmaxf(fmaxf(a.x, b.x), c.x), fmaxf(fmaxf(a.y, b.y), c.y), fmaxf(fmaxf(a.z, b.z), c.z), fmaxf(fmaxf(a.w, b.w), c.w));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

typedef void (*Stem32Fn)(Stem32Params, ActCfg, const float*, const float*, const float4*, float*);

struct Stem32Plan : ConvPlanBase {
    Stem32Params p;
    ActCfg ac;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    dim3 grid;
    size_t ldsBytes = 0;
    Stem32Fn kernel = nullptr;

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "conv2d: expects 1 input, got %d", nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.H && x->w == p.W && x->c == p.IC && x->dtype == SNNHIP_F32, "conv2d: input dims %dx%dx%dx%d (dtype %d) != plan %dx%dx%dx%d fp32",
                       x->n, x->h, x->w, x->c, x->dtype, p.N, p.H, p.W, p.IC);
        const int eoh = p.poolOH ? p.poolOH : p.OH, eow = p.poolOH ? p.poolOW : p.OW;
        SNNHIP_REQUIRE(out->n == p.N && out->h == eoh && out->w == eow && out->c == p.OC && out->dtype == SNNHIP_F32, "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d",
                       out->n, out->h, out->w, out->c, p.N, eoh, eow, p.OC);
        SNNHIP_LAUNCH(kernel, grid, dim3(256), ldsBytes, ctx->stream, p, ac, x->data, d_w, reinterpret_cast<const float4*>(d_epi), out->data);
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

template <int K, int S>
Stem32Fn pick_stem32(bool simple, int nr) {
    if (nr == 2) return simple ? conv2d_stem32_kernel<K, S, 2, true> : conv2d_stem32_kernel<K, S, 2, false>;
    return simple ? conv2d_stem32_kernel<K, S, 4, true> : conv2d_stem32_kernel<K, S, 4, false>;
}

} // namespace

// fp32, IC <= 4, OC % 32 == 0, (k, stride) in {(3, 1), (3, 2), (7, 2)}, no fused Pad / residual; SNNHIP_CONV_STEM=0 leaves the layer to conv2d_mfma's
// tap-pair mode
int make_conv2d_stem32_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    if (g.dtype != SNNHIP_F32 || g.kh != g.kw || g.sh != g.sw || g.IC > 4 || g.OC % 32 != 0) return SNNHIP_E_UNSUPPORTED;
    if (g.addAct >= 0 || g.preMode || g.normShift || g.act == SNNHIP_ACT_SILU_QUIRK) return SNNHIP_E_UNSUPPORTED; // (the quirk couples 4 adjacent pixels)
    if (const char* e = snnhip::option("SNNHIP_CONV_STEM"))
        if (atoi(e) == 0) return SNNHIP_E_UNSUPPORTED;
    if (const char* f = snnhip::option("SNNHIP_CONV"))
        if (strcmp(f, "stem") != 0) return SNNHIP_E_UNSUPPORTED; // another kernel is being forced
    const bool simple = act_is_simple(g.act);
    Stem32Fn fn = nullptr;
    int K = g.kh, S = g.sh;
    // 16-row tiles (4 rows per wave) unless they leave the chip under-filled: a single 224x224 image is 196 blocks on 256 CUs and the launch takes one
    // block's latency -- with 8-row tiles it is 392 blocks of half the work (BASELINE configs[0]: 8.5 -> 6 us)
    const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
    const long blocks16 = static_cast<long>(g.N) * up_div(g.OH, 16) * up_div(g.OW, kTW) * (g.OC / 32);
    int NR = blocks16 < 2L * cus ? 2 : 4;
    if (const char* e = snnhip::option("SNNHIP_STEM_ROWS")) NR = atoi(e) == 2 ? 2 : 4;
    const int kTH = 4 * NR;
    if (K == 3 && S == 1) fn = pick_stem32<3, 1>(simple, NR);
    if (K == 3 && S == 2) fn = pick_stem32<3, 2>(simple, NR);
    if (K == 7 && S == 2) fn = pick_stem32<7, 2>(simple, NR);
    // the 7x7 stride-2 RGB stem: dense (tap, channel) packing of the reduction, 74 instead of 98 MFMAs per row tile (SNNHIP_STEM_DENSE=0: the general form)
    bool dense = K == 7 && S == 2 && g.IC == 3;
    if (const char* e = snnhip::option("SNNHIP_STEM_DENSE")) dense = dense && atoi(e) != 0;
    if (dense) fn = NR == 2 ? (simple ? conv2d_stem32_dense_kernel<2, true> : conv2d_stem32_dense_kernel<2, false>)
                            : (simple ? conv2d_stem32_dense_kernel<4, true> : conv2d_stem32_dense_kernel<4, false>);
    if (!fn) return SNNHIP_E_UNSUPPORTED;
    if (static_cast<double>(g.N) * g.OH * g.OW * g.OC >= 2147483647.0 * 2) return SNNHIP_E_UNSUPPORTED;

    Stem32Params p{};
    p.N = g.N; p.H = g.H; p.W = g.W; p.IC = g.IC; p.OC = g.OC; p.padx = g.padx; p.pady = g.pady; p.padMode = g.padMode; p.useBN = g.useBN;
    p.OH = g.OH; p.OW = g.OW;
    p.tilesX = up_div(g.OW, kTW); p.tilesY = up_div(g.OH, kTH);
    const int IN_H = (kTH - 1) * S + K, IN_W = (kTW - 1) * S + K;
    (void) IN_W;
    const size_t lds = dense ? (static_cast<size_t>(3) * IN_H * 2 * kDensePW + 4 * 32 * kOutPitch + 32 * 4) * sizeof(float) // [3][IN_H][2][kDensePW] floats, scratch, epilogue table
                             : (static_cast<size_t>(IN_H) * (2 * (S == 2 ? 2 : 1) * plane_w(K, S) * 2) + 4 * 32 * kOutPitch) * sizeof(float); // [IN_H][2][parity planes][kPlaneW] pairs
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
        set_error("conv2d_stem32: hipFuncSetAttribute(%zu) failed", lds);
        return SNNHIP_E_HIP;
    }

    auto* plan = new Stem32Plan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * K * K);
    plan->epi4 = epi4;
    plan->p = p;
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->kernel = fn;
    plan->ldsBytes = lds;
    plan->grid = dim3(static_cast<unsigned>(p.tilesX) * p.tilesY * g.N, g.OC / 32, 1);

    // weights: Wp[oc block][step = 2 (fy K + fx) + m][lane = 32 hh + o] = W[32 b + o][ic = 2m + hh][fy][fx]
    const int NW = 2 * K * K, ocb = g.OC / 32;
    std::vector<float> wpk(static_cast<size_t>(ocb) * NW * 64, 0.0f);
    for (int b = 0; b < ocb; ++b)
        for (int t = 0; t < K * K; ++t)
            for (int m = 0; m < 2; ++m)
                for (int hh = 0; hh < 2; ++hh) {
                    const int ic = 2 * m + hh;
                    if (ic >= g.IC) continue;
                    for (int o = 0; o < 32; ++o)
                        wpk[(static_cast<size_t>(b) * NW + 2 * t + m) * 64 + 32 * hh + o] = w_oihw[(static_cast<size_t>(b * 32 + o) * g.IC + ic) * K * K + t];
                }
    if (dense) { // Wp[oc block][k][lane = 32 hh + o] = W_flat[32 b + o][e = 2k + hh], e = (fy 7 + fx) 3 + ch, zero behind e = 146
        wpk.assign(static_cast<size_t>(ocb) * kDenseSteps * 64, 0.0f);
        for (int b = 0; b < ocb; ++b)
            for (int k = 0; k < kDenseSteps; ++k)
                for (int hh = 0; hh < 2; ++hh) {
                    const int e = 2 * k + hh;
                    if (e >= 147) continue;
                    const int ch = e % 3, t = e / 3;
                    for (int o = 0; o < 32; ++o) wpk[(static_cast<size_t>(b) * kDenseSteps + k) * 64 + 32 * hh + o] = w_oihw[(static_cast<size_t>(b * 32 + o) * 3 + ch) * 49 + t];
                }
    }
    int rc = plan->upload(wpk.data(), wpk.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epi4.data(), epi4.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = g.N; plan->inDims[1] = g.H; plan->inDims[2] = g.W; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->dtype = SNNHIP_F32;
    plan->flops = 2.0 * K * K * g.IC * g.OC * static_cast<double>(g.OH) * g.OW * g.N;
    plan->bytes = 4.0 * (static_cast<double>(g.N) * g.H * g.W * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC + static_cast<double>(g.OC) * g.IC * K * K);
    char buf[256];
    snprintf(buf, sizeof(buf), "conv2d_mfma_stem_f32_32x32x2 k=%dx%d s=%d ic=%d oc=%d tile=%dx%dpx x 32oc (%s, weights in registers) lds=%zuB", K, K, S, g.IC, g.OC, kTH, kTW,
             dense ? "dense (tap, channel) K: 74 MFMAs per row tile" : "2 MFMAs per tap", lds);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

// Chain rule J: the 7x7 stride-2 RGB stem -> MaxPooling2D 3x3 stride 2 as one launch (conv2d_stem32_dense_kernel<4, ., true>)
int make_conv2d_stem32_pool_plan(snnhip_ctx* ctx, snnhip_plan* stemPlan, snnhip_plan* poolPlan, snnhip_plan** out) {
    if (snnhip::option("SNNHIP_NO_STEM_POOL_FUSION")) return SNNHIP_E_UNSUPPORTED;
    auto* cs = dynamic_cast<Stem32Plan*>(stemPlan);
    snnhip_pool2d_desc pd;
    if (!cs || !pool2d_plan_desc(poolPlan, &pd)) return SNNHIP_E_UNSUPPORTED;
    const ConvGeom& g = cs->g;
    if (cs->desc.find("dense (tap, channel) K") == std::string::npos) return SNNHIP_E_UNSUPPORTED; // the dense 7x7 stride-2 RGB form only
    if (pd.type != SNNHIP_POOL_MAX || pd.kh != 3 || pd.kw != 3 || pd.sh != 2 || pd.sw != 2 || pd.padT != 0 || pd.padL != 0) return SNNHIP_E_UNSUPPORTED;
    if (pd.N != g.N || pd.H != g.OH || pd.W != g.OW || pd.C != g.OC) return SNNHIP_E_UNSUPPORTED;
    // every pooled pixel's window must start inside the convolution's output (rows 2 py, columns 2 px): true for the reference's output-size rule
    if (2 * (pd.OH - 1) >= g.OH || 2 * (pd.OW - 1) >= g.OW) return SNNHIP_E_UNSUPPORTED;
    const bool simple = act_is_simple(g.act);
    Stem32Fn fn = simple ? conv2d_stem32_dense_kernel<4, true, true> : conv2d_stem32_dense_kernel<4, false, true>;
    Stem32Params p = cs->p;
    p.poolOH = pd.OH;
    p.poolOW = pd.OW;
    p.tilesX = up_div(pd.OW, 15);
    p.tilesY = up_div(pd.OH, 7);
    constexpr int IN_H = 15 * 2 + 7;
    const size_t lds = (static_cast<size_t>(3) * IN_H * 2 * kDensePW + 4 * 32 * kOutPitch + 32 * 4) * sizeof(float);
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
        set_error("conv2d_stem32+pool: hipFuncSetAttribute(%zu) failed", lds);
        return SNNHIP_E_HIP;
    }
    auto* plan = new Stem32Plan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw = cs->w_oihw;
    plan->epi4 = cs->epi4;
    plan->p = p;
    plan->ac = cs->ac;
    plan->kernel = fn;
    plan->ldsBytes = lds;
    plan->grid = dim3(static_cast<unsigned>(p.tilesX) * p.tilesY * g.N, g.OC / 32, 1);
    // the packed weights and the epilogue table of the 16-row-tile dense form are what this kernel reads: re-packed here (the borrowed plan may use 8-row tiles)
    const int ocb = g.OC / 32;
    std::vector<float> wpk(static_cast<size_t>(ocb) * kDenseSteps * 64, 0.0f);
    for (int b = 0; b < ocb; ++b)
        for (int k = 0; k < kDenseSteps; ++k)
            for (int hh = 0; hh < 2; ++hh) {
                const int e = 2 * k + hh;
                if (e >= 147) continue;
                const int ch = e % 3, t = e / 3;
                for (int o = 0; o < 32; ++o) wpk[(static_cast<size_t>(b) * kDenseSteps + k) * 64 + 32 * hh + o] = cs->w_oihw[(static_cast<size_t>(b * 32 + o) * 3 + ch) * 49 + t];
            }
    int rc = plan->upload(wpk.data(), wpk.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(cs->epi4.data(), cs->epi4.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    memcpy(plan->inDims, stemPlan->inDims, sizeof(plan->inDims));
    memcpy(plan->outDims, poolPlan->outDims, sizeof(plan->outDims));
    plan->dtype = SNNHIP_F32;
    plan->flops = stemPlan->flops + poolPlan->flops;
    plan->bytes = stemPlan->bytes + poolPlan->bytes; // unfused accounting of the two layers it replaces
    // what the one launch moves: the image, the pooled tensor, the weights (the stem's output never reaches memory)
    plan->kernelBytes = 4.0 * (static_cast<double>(g.N) * g.H * g.W * g.IC + static_cast<double>(plan->outDims[0]) * plan->outDims[1] * plan->outDims[2] * plan->outDims[3] +
                               static_cast<double>(g.OC) * g.IC * 49);
    char buf[384];
    snprintf(buf, sizeof(buf), "conv2d_mfma_stem_f32_32x32x2 k=7x7 s=2 ic=3 oc=%d tile=16x32px x 32oc (dense (tap, channel) K: 74 MFMAs per row tile, weights in registers) "
             "+maxpool3x3/2 in the epilogue (7x15 pooled px per block) lds=%zuB hbm_bytes=%.6g", g.OC, lds, plan->kernelBytes);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
