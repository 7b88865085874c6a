// espcn_stream.hip -- the whole ESPCN-shaped chain in ONE kernel, fp32:
//     conv KxK (1 -> 16) + act   ->  conv 3x3 (16 -> 16) + act  ->  conv 3x3 (16 -> 4) + act  ->  depth-to-space(2) + tanh
// replacing four render passes of the reference (3x shadertemplate_vk_conv2d.comp:148-347 + shadertemplate_vk_subpixel.comp:43-71).
//
// Design ("wave-private row streaming"):
//   * one wave64 = one workgroup = one column strip (60 output columns) x one row segment (R rows); no barriers,
//     no inter-wave communication, 11 KB of LDS per wave, all 2048 waves of a 1080p frame are resident at once
//   * the wave marches down its rows.  Per row it does
//        MFMA : conv1 of one row (K = 25 taps padded to 28, 7 x 4 v_mfma_f32_16x16x4_f32), bias/act   -> LDS row buffer
//        MFMA : that conv1 row's contribution to THREE conv2 output rows (rolling accumulators):
//               D[oc][px] += W2[fy][fx][oc][ic] * c1[row][px+fx][ic]  as v_mfma_f32_16x16x4_f32, 144 per row,
//               conv2 weights resident in 36 VGPRs, B operand = 12 ds_read_b128 per row
//        VALU : the finished conv2 row -> bias/act -> LDS row buffer -> its contribution to THREE conv3 output rows
//               (12 rolling partial sums per lane), the finished conv3 row -> act -> tanh -> 2x2 pixel-shuffle store
//     so nothing but the 1-channel input and the 4K output ever touches HBM (20 B/px instead of 308 B/px unfused),
//     and the kernel is bound by the fp32 matrix pipe (conv1 + conv2 = 82% of the flops) with conv3, the epilogues and
//     the pixel shuffle on the otherwise idle VALU: two co-resident waves per SIMD alternate between the two pipes.
//     (conv1 on the VALU with scalar weights was tried first: 400 wave-uniform weights per row blow the SGPR file.)
//   * vertical taps are handled by accumulator rotation instead of row rings, so LDS holds ONE row of each tensor.
// fp32 MFMA is a plain fp32 fma chain: results stay within 1e-4 of the CPU oracle like the VALU kernels.
#include "epilogue.h"
#include "snnhip_internal.h"

#ifndef SNNHIP_STREAM_STAMP
#define SNNHIP_STREAM_STAMP(k)
#endif

namespace snnhip {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// A workgroup is ONE wave: LDS instructions of a wave execute in program order, so cross-lane hand-offs through LDS
// need no s_barrier -- only a guarantee that the compiler keeps the ds_write / ds_read program order.
#define SNNHIP_WAVE_SYNC()                                       \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

constexpr int STRIP = 60; // conv3 output columns per wave; conv2 computes 64 (4 MFMA groups), conv1 exactly 64 lanes

struct StreamParams {
    int N, H, W;
    int numStrips, numSegs, segRows;
    ActCfg act1, act2, act3;
};

// WPB waves per workgroup (each wave still fully independent): a workgroup's waves are spread over the CU's four SIMDs,
// which single-wave workgroups are not guaranteed to be.
constexpr int WPB = 4;

template <int K1, bool SIMPLE>
__global__ __launch_bounds__(64 * WPB, 2) void espcn_stream_kernel(StreamParams p, const float* __restrict__ x, const float* __restrict__ wA1,
                                                                   const float* __restrict__ ep1, const float* __restrict__ wA2,
                                                                   const float* __restrict__ ep2, const float* __restrict__ w3r,
                                                                   const float* __restrict__ ep3, float* __restrict__ y) {
    constexpr int P1 = K1 / 2;
    constexpr int INW = 64 + 2 * P1;       // input columns per row: X0-2-P1 .. X0+61+P1
    constexpr int INROWS = K1;             // ring depth
    constexpr int C2P = 20;                // c2 row-buffer pitch (floats): conflict-free b128 reads for consecutive lanes
    constexpr int KS1 = (K1 * K1 + 3) / 4; // MFMA K-steps of conv1 (K = taps, padded to a multiple of 4)
    constexpr int INSZ = (INROWS * INW + 3) & ~3;
    constexpr int PERWAVE = INSZ + 66 * 16 + 66 * C2P;
    __shared__ __attribute__((aligned(16))) float smem[WPB * PERWAVE];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* s_in = smem + wv * PERWAVE;
    float* s_c1 = s_in + INSZ;
    float* s_c2 = s_c1 + 66 * 16;

    const int lane = threadIdx.x & 63;
    const int px = lane & 15, g = lane >> 4;

    for (int i = lane; i < 66 * 16; i += 64) s_c1[i] = 0.0f; // includes the 2 pad pixels read only by discarded lanes
    for (int i = lane; i < 66 * C2P; i += 64) s_c2[i] = 0.0f; // "row Y0-4": contributes nothing to the first conv3 step
    SNNHIP_WAVE_SYNC();

    int b = blockIdx.x * WPB + wv;
    if (b >= p.numStrips * p.numSegs * p.N) return;
    const int strip = b % p.numStrips;
    b /= p.numStrips;
    const int seg = b % p.numSegs;
    const int n = b / p.numSegs;
    const int X0 = strip * STRIP;
    const int Y0 = seg * p.segRows;
    const int Y1 = min(Y0 + p.segRows, p.H); // output rows [Y0, Y1)
    const float* xn = x + static_cast<size_t>(n) * p.H * p.W;
    float* yn = y + static_cast<size_t>(n) * (2 * p.H) * (2 * p.W);

    // conv2 weights: MFMA A operand, lane-packed by the host: a2[(fy*3+fx)*4 + j][lane] = W2[oc = lane&15][ic = 4*(lane>>4)+j][fy][fx]
    float a2[36];
#pragma unroll
    for (int t = 0; t < 36; ++t) a2[t] = wA2[t * 64 + lane];
    // conv1 weights: a1[s][lane] = W1[oc = lane&15][tap = 4s + (lane>>4)] (0 beyond the last tap)
    float a1[KS1];
#pragma unroll
    for (int s = 0; s < KS1; ++s) a1[s] = wA1[s * 64 + lane];
    float sc1[4], sh1[4], sc2[4], sh2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sc1[r] = ep1[(4 * g + r) * 2];
        sh1[r] = ep1[(4 * g + r) * 2 + 1];
        sc2[r] = ep2[(4 * g + r) * 2];
        sh2[r] = ep2[(4 * g + r) * 2 + 1];
    }
    float sc3[4], sh3[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc3[k] = ep3[2 * k];
        sh3[k] = ep3[2 * k + 1];
    }
    // this lane's conv1 tap for K-step s: t = 4s+g -> (fy, fx); taps >= K1*K1 have zero weights, their B value is forced to 0
    int tapFy[KS1], tapOff[KS1];
    bool tapOk[KS1];
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
        int t = 4 * s + g;
        tapOk[s] = t < K1 * K1;
        if (!tapOk[s]) t = K1 * K1 - 1;
        tapFy[s] = t / K1;
        tapOff[s] = (t - tapFy[s] * K1) + px;
    }

    // ---- input ring: row r lives in slot (r mod K1)
    auto load_row = [&](int r, float& v0, float& v1) {
        v0 = 0.0f;
        v1 = 0.0f;
        if (r >= 0 && r < p.H) {
            const int c0 = X0 - 2 - P1 + lane;
            if (c0 >= 0 && c0 < p.W) v0 = xn[static_cast<size_t>(r) * p.W + c0];
            const int c1 = c0 + 64;
            if (lane < 2 * P1 && c1 < p.W) v1 = xn[static_cast<size_t>(r) * p.W + c1]; // c1 >= 60 > 0 always
        }
    };
    auto store_row = [&](int r, float v0, float v1) {
        int slot = r % K1;
        if (slot < 0) slot += K1;
        s_in[slot * INW + lane] = v0;
        if (lane < 2 * P1) s_in[slot * INW + 64 + lane] = v1;
    };

    const int jFirst = Y0 - 2; // first conv1 row this wave needs (for conv2 row Y0-1)
    const int jLast = Y1 + 1;  // last conv1 row (for conv2 row Y1)
    for (int r = jFirst - P1; r <= jFirst + P1; ++r) { // conv1 row j reads input rows j-P1 .. j+P1
        float v0, v1;
        load_row(r, v0, v1);
        store_row(r, v0, v1);
    }
    SNNHIP_WAVE_SYNC();

    f32x4 accA[4], accB[4], accC[4]; // conv2 rows j-1, j, j+1 while processing conv1 row j
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
        accA[gi] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        accB[gi] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        accC[gi] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    float pA[4] = {0, 0, 0, 0}, pB[4] = {0, 0, 0, 0}, pC[4] = {0, 0, 0, 0}; // conv3 rows q-1, q, q+1 while consuming conv2 row q

    const int outCol = X0 + lane;
    const bool outColOk = lane < STRIP && outCol < p.W;

    // conv3 step: consumes the conv2 row sitting in s_c2 (row q, zeros when q is outside the image or not needed),
    // adds its contribution to output rows q+1 (dy=0), q (dy=1), q-1 (dy=2); row q-1 is then complete and stored.
    // Weights come from LDS as wave-uniform b128 reads: s_w3[((dx*4+qd)*4+i)*12 + dy*4 + o].
    // conv3 weights: 576 wave-uniform scalars.  They live lane-distributed in 9 VGPRs (weight k = lane k%64 of register
    // k/64) and are broadcast with v_readlane_b32 (-> SGPR operand of v_fmac): no memory latency on the VALU side, which
    // is what lets conv3 hide under the conv2 MFMAs.  (Scalar loads blow the SGPR budget, LDS broadcasts add a
    // ~120-cycle latency chain per 4 weights; both were measured slower.)
    float w3v[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) w3v[r] = w3r[r * 64 + lane];
    auto w3 = [&](int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w3v[k >> 6]), k & 63)); };
    auto conv3_substep = [&](int dq) { // one (dx, input-channel quad): 1 activation b128 + 48 readlane + 48 FMAs
        const int dx = dq >> 2, qd = dq & 3;
        const float4 xv = *reinterpret_cast<const float4*>(s_c2 + (lane + dx) * C2P + qd * 4);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = (dq * 4 + i) * 12;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                pC[o] = fmaf(xs[i], w3(k + 0 + o), pC[o]);
                pB[o] = fmaf(xs[i], w3(k + 4 + o), pB[o]);
                pA[o] = fmaf(xs[i], w3(k + 8 + o), pA[o]);
            }
        }
        // pin the 48 FMAs here: without it the optimiser sinks them to the end of the row
        asm volatile("" : "+v"(pA[0]), "+v"(pA[1]), "+v"(pA[2]), "+v"(pA[3]), "+v"(pB[0]), "+v"(pB[1]), "+v"(pB[2]), "+v"(pB[3]), "+v"(pC[0]),
                     "+v"(pC[1]), "+v"(pC[2]), "+v"(pC[3]));
    };
    auto conv3_finish = [&](int q) { // output row q-1 is complete: act, tanh, 2x2 pixel shuffle; rotate the partial sums
        const int oy = q - 1;
        if (oy >= Y0 && oy < Y1 && outColOk) {
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fast_tanh(apply_act<SIMPLE>(p.act3, fmaf(pA[k], sc3[k], sh3[k]), 0.0f));
            *reinterpret_cast<float2*>(yn + static_cast<size_t>(2 * oy) * (2 * p.W) + 2 * outCol) = make_float2(o[0], o[1]);
            *reinterpret_cast<float2*>(yn + static_cast<size_t>(2 * oy + 1) * (2 * p.W) + 2 * outCol) = make_float2(o[2], o[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pA[k] = pB[k];
            pB[k] = pC[k];
            pC[k] = 0.0f;
        }
    };

    for (int j = jFirst; j <= jLast; ++j) {
        SNNHIP_STREAM_STAMP(0);
        // ---- prefetch input row j+P1+1 (first needed by conv1 row j+1) into registers; it replaces row j-P1 in the
        //      ring once conv1(j) has read it, so the HBM/L2 latency hides under this row's work
        float nv0, nv1;
        load_row(j + P1 + 1, nv0, nv1);

        // ---- conv1 row j on the matrix pipe: D[oc][px] = W1[oc][tap] * in[row j-P1+fy][col px+fx], 4 groups of 16 columns.
        //      Rows outside the image are conv2's zero padding: the epilogue masks them to 0 (no branches in this loop).
        const bool rowIn = j >= 0 && j < p.H;
        {
            f32x4 acc1[4];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) acc1[gi] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            int slot0 = (j - P1) % K1;
            if (slot0 < 0) slot0 += K1;
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                int slot = slot0 + tapFy[s];
                slot = slot >= K1 ? slot - K1 : slot;
                const float* src = s_in + slot * INW + tapOff[s];
                float bv[4];
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    bv[gi] = src[gi * 16];
                    bv[gi] = tapOk[s] ? bv[gi] : 0.0f;
                }
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) acc1[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bv[gi], acc1[gi], 0, 0, 0);
            }
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const int i = gi * 16 + px; // c1 buffer index, column X0-2+i
                const int col = X0 - 2 + i;
                const bool in = rowIn && col >= 0 && col < p.W;
                float4 o;
                o.x = apply_act<SIMPLE>(p.act1, fmaf(acc1[gi][0], sc1[0], sh1[0]), 0.0f);
                o.y = apply_act<SIMPLE>(p.act1, fmaf(acc1[gi][1], sc1[1], sh1[1]), 0.0f);
                o.z = apply_act<SIMPLE>(p.act1, fmaf(acc1[gi][2], sc1[2], sh1[2]), 0.0f);
                o.w = apply_act<SIMPLE>(p.act1, fmaf(acc1[gi][3], sc1[3], sh1[3]), 0.0f);
                o.x = in ? o.x : 0.0f;
                o.y = in ? o.y : 0.0f;
                o.z = in ? o.z : 0.0f;
                o.w = in ? o.w : 0.0f;
                const int slot4 = g ^ (((i >> 2) & 1) << 1);
                *reinterpret_cast<float4*>(s_c1 + i * 16 + slot4 * 4) = o;
            }
        }
        SNNHIP_WAVE_SYNC();
        store_row(j + P1 + 1, nv0, nv1);
        SNNHIP_STREAM_STAMP(1);

        // ---- conv2 (MFMA): contributions of conv1 row j to conv2 rows j+1 (fy=0), j (fy=1), j-1 (fy=2); in the same
        //      scheduling region the VALU runs conv3 on the PREVIOUS conv2 row (s_c2), so one wave feeds both pipes
        {
            float4 bv[3][4];
#pragma unroll
            for (int fx = 0; fx < 3; ++fx)
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const int cc = gi * 16 + px + fx; // c1 buffer index
                    const int slot = g ^ (((cc >> 2) & 1) << 1);
                    bv[fx][gi] = *reinterpret_cast<const float4*>(s_c1 + cc * 16 + slot * 4);
                }
#pragma unroll
            for (int fx = 0; fx < 3; ++fx) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) accC[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[(0 * 3 + fx) * 4 + jj], bv[fx][gi][jj], accC[gi], 0, 0, 0);
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) accB[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[(1 * 3 + fx) * 4 + jj], bv[fx][gi][jj], accB[gi], 0, 0, 0);
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) accA[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[(2 * 3 + fx) * 4 + jj], bv[fx][gi][jj], accA[gi], 0, 0, 0);
                    // one twelfth of conv3 (previous conv2 row, in s_c2) rides in the shadow of these 12 MFMAs; the
                    // scheduling barrier keeps the compiler from hoisting all 144 weight reads (it spills otherwise)
                    conv3_substep(fx * 4 + jj);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        conv3_finish(j - 2); // s_c2 held conv2 row j-2 (written at the end of the previous iteration)
        SNNHIP_STREAM_STAMP(2);

        // ---- conv2 row r = j-1 is complete: bias/act, zero outside the image or when no output row needs it -> s_c2
        {
            const int r = j - 1;
            const bool rOk = r >= Y0 - 1 && r <= Y1 && r >= 0 && r < p.H;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const int i = gi * 16 + px; // c2 buffer index, column X0-1+i
                const int col = X0 - 1 + i;
                const bool in = rOk && col >= 0 && col < p.W;
                float4 o;
                o.x = apply_act<SIMPLE>(p.act2, fmaf(accA[gi][0], sc2[0], sh2[0]), 0.0f);
                o.y = apply_act<SIMPLE>(p.act2, fmaf(accA[gi][1], sc2[1], sh2[1]), 0.0f);
                o.z = apply_act<SIMPLE>(p.act2, fmaf(accA[gi][2], sc2[2], sh2[2]), 0.0f);
                o.w = apply_act<SIMPLE>(p.act2, fmaf(accA[gi][3], sc2[3], sh2[3]), 0.0f);
                o.x = in ? o.x : 0.0f;
                o.y = in ? o.y : 0.0f;
                o.z = in ? o.z : 0.0f;
                o.w = in ? o.w : 0.0f;
                *reinterpret_cast<float4*>(s_c2 + i * C2P + g * 4) = o;
            }
        }
        // ---- rotate conv2 accumulators
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            accA[gi] = accB[gi];
            accB[gi] = accC[gi];
            accC[gi] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        SNNHIP_WAVE_SYNC(); // next iteration overwrites s_c1 and reads s_c2
        SNNHIP_STREAM_STAMP(3);
    }
#pragma unroll
    for (int dq = 0; dq < 12; ++dq) {
        conv3_substep(dq);
        __builtin_amdgcn_sched_barrier(0);
    }
    conv3_finish(jLast - 1); // conv2 row Y1 -> completes output row Y1-1
}

} // namespace

// Rule C of snnhip_chain_plan_create: the complete ESPCN pattern -> one launch.
struct StreamStep {
    StreamParams sp;
    int k1;
    bool simple;
};

int espcn_stream_launch(hipStream_t stream, const void* step, const float* x, const float* wA1, const float* ep1, const float* wA2, const float* ep2,
                        const float* w3r, const float* ep3, float* y) {
    const StreamStep& s = *static_cast<const StreamStep*>(step);
    dim3 grid(up_div(s.sp.numStrips * s.sp.numSegs * s.sp.N, WPB));
#define SNNHIP_LAUNCH_S(K, S) SNNHIP_LAUNCH((espcn_stream_kernel<K, S>), grid, dim3(64 * WPB), 0, stream, s.sp, x, wA1, ep1, wA2, ep2, w3r, ep3, y)
    if (s.k1 == 5) {
        if (s.simple) SNNHIP_LAUNCH_S(5, true); else SNNHIP_LAUNCH_S(5, false);
    } else {
        if (s.simple) SNNHIP_LAUNCH_S(3, true); else SNNHIP_LAUNCH_S(3, false);
    }
#undef SNNHIP_LAUNCH_S
    SNNHIP_CHECK_HIP(hipGetLastError());
    return SNNHIP_OK;
}

size_t espcn_stream_step_size() { return sizeof(StreamStep); }

void espcn_stream_configure(void* step, int N, int H, int W, int k1, int act1, float leaky1, int act2, float leaky2, int act3, float leaky3,
                            int computeUnits) {
    StreamStep& s = *static_cast<StreamStep*>(step);
    s.k1 = k1;
    s.simple = act_is_simple(act1) && act_is_simple(act2) && act_is_simple(act3);
    s.sp.N = N;
    s.sp.H = H;
    s.sp.W = W;
    s.sp.numStrips = up_div(W, STRIP);
    // aim at 8 resident waves per CU (2 per SIMD) over the whole grid; at least 8 rows per segment
    int segs = (computeUnits * 8) / (s.sp.numStrips * N);
    if (segs < 1) segs = 1;
    int rows = up_div(H, segs);
    if (rows < 8) rows = H < 8 ? H : 8;
    s.sp.segRows = rows;
    s.sp.numSegs = up_div(H, rows);
    s.sp.act1 = make_act_cfg(act1, leaky1);
    s.sp.act2 = make_act_cfg(act2, leaky2);
    s.sp.act3 = make_act_cfg(act3, leaky3);
}

void espcn_stream_describe(const void* step, char* buf, size_t n) {
    const StreamStep& s = *static_cast<const StreamStep*>(step);
    snprintf(buf, n, "fused[conv%dx%d(1->16)+conv3x3(16->16)+conv3x3(16->4)+depth_to_space(2)+tanh] stream mfma_f32_16x16x4+valu strip=%d rows/seg=%d waves=%d",
             s.k1, s.k1, STRIP, s.sp.segRows, s.sp.numStrips * s.sp.numSegs * s.sp.N);
}

} // namespace snnhip
