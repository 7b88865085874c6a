// conv2d_ksplit.hip -- fp32 Conv2D as an implicit GEMM whose operands never pass through LDS and whose K axis is split over the WAVES of a
// block (round 6): the fp32 3x3 stride-2 layers of the ResNet-18 body (BASELINE configs[2]: 64 -> 128 @56, 128 -> 256 @28, 256 -> 512 @14 at
// batch 32), which are small GEMMs with a deep K (M = 25 088 / 6 272 / 1 568 output pixels, K = 576 / 1 152 / 2 304).  conv2d_mfma_kernel runs
// them with split-K over blockIdx.z + a reduce / epilogue launch (partial sums through HBM, 0.33 - 0.44 of the fp32 MFMA peak, three extra
// launches per step); here the partial tiles are summed through LDS inside the block and there is ONE launch.
//
// Same operator contract as conv2d_mfma.hip / conv2d_generic.hip (shadertemplate_vk_conv2d.comp:148-347: zero padding by clipped taps, bias ->
// BN -> activation epilogue).  Not offered: fused Pad / UpSampling / InstanceNorm in front, fused Add behind (those shapes keep conv2d_mfma).
//
// Work decomposition
//   block  = KS waves, ONE output tile of 32 MI pixels (rows of the GEMM: consecutive pixels of the NHWC output, images included) x 64 channels
//   wave k = the WHOLE tile over its share [k T / KS, (k + 1) T / KS) of the T = kh kw IC / 16 K-iterations (tap-major, 16 channels each):
//            MI x 2 accumulators of v_mfma_f32_32x32x2_f32
//   A operand (activations): stride 2 makes every input pixel a tap of 2.25 outputs on average -- there is nothing an LDS halo tile would save
//            that the L2 does not -- so lane (row = l % 32, h = l / 32) loads the 32 bytes x[pixel(row) + tap][16 c + 8 h ..] of an iteration
//            straight into the MFMA's A layout (two 16-byte buffer loads: the eight components are the eight K steps; a lane pair covers a
//            64-byte sector).  A tap outside the image is a buffer offset beyond the descriptor's range: the hardware returns zeros, no branch.
//   B operand (weights): pre-packed [oc tile][iteration][n tile][half][lane] float4 in the same K permutation, 4 KB per iteration read as four
//            perfectly coalesced 1 KB loads; the blocks of a launch walk the pixel tiles first, so every resident block reads the same 0.3 - 0.6
//            MB weight column out of the L2.
//   loads are DEPTH iterations ahead in registers (no barrier, no LDS in the K loop), 16 MI MFMAs per iteration
//   reduction: every wave writes its accumulators to LDS as [k][quad][lane] float4 (conflict-free), ONE barrier, then wave k sums quads
//            [k G / KS, (k + 1) G / KS) of all KS partial tiles in a fixed order (deterministic), applies the epilogue and stores: a half wave's
//            lanes are 32 channels of one pixel (128 contiguous bytes per store instruction)
//   bound: fp32 MFMA (2 * 9 * IC * OC FLOP per output pixel); the grid is sized so that every SIMD gets several short waves (balance) instead of
//   one long one.
#include <cmath>
#include <cstring>
#include <vector>

#include "epilogue.h"
#include "snnhip_internal.h"

#ifndef SNNHIP_KS_ABL
#define SNNHIP_KS_ABL 0 // ablation builds only (tools/exp_one.sh; results are wrong by construction): 1 no weight loads, 2 no activation loads, 4 no reduction / epilogue
#endif

namespace snnhip {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct KsParams {
    int N, H, W, IC, OC, OH, OW, kh, kw, sh, sw, padx, pady;
    int M;        // N * OH * OW output pixels (GEMM rows)
    int T;        // K iterations: kh * kw * cpt
    int cpt;      // 16-channel chunks per tap
    int mTiles;   // ceil(M / (32 MI))
    int useBN;
    unsigned xBytes, wBytes;
};

constexpr unsigned kOutOfRange = 0x80000000u; // a byte offset no descriptor of this kernel covers (tensors < 2 GiB): the load returns zeros

// one block's work: output tile `blk` of the problem (p, x, wpk, epi, y); red = the block's LDS, [KS][G][64] float4
template <bool SIMPLE, int MI, int KS, int DEPTH>
__device__ __forceinline__ void ksplit_tile(const KsParams& p, const ActCfg& ac, const float* __restrict__ x, const float4* __restrict__ wpk,
                                            const float4* __restrict__ epi, float* __restrict__ y, const int blk, f32x4* const red) {
    constexpr int G = 8 * MI;               // accumulator quads (float4) per lane
    constexpr int GW = (G + KS - 1) / KS;   // quads a wave owns in the reduction (quad g belongs to wave g % KS)
    static_assert(KS >= 1 && KS <= G, "at most one wave per accumulator quad");
    const int lane = threadIdx.x & 63, k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // (k in a scalar register: the K range, the tap walk and the loop are wave-uniform)
    const int r = lane & 31, h = lane >> 5;
    const int mTile = blk % p.mTiles, ocTile = blk / p.mTiles;

    const __amdgpu_buffer_rsrc_t xRsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, static_cast<int>(p.xBytes), 0x00020000);

    // ---- this lane's MI output pixels: where their top-left taps sit
    int base[MI], iy0[MI], ix0[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mTile * (32 * MI) + 32 * i + r;
        const int mm = m < p.M ? m : 0;
        const int n = mm / (p.OH * p.OW), rem = mm - n * (p.OH * p.OW);
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        iy0[i] = m < p.M ? oy * p.sh - p.pady : -(1 << 20); // rows beyond M: every tap out of range
        ix0[i] = ox * p.sw - p.padx;
        base[i] = (((n * p.H + iy0[i]) * p.W + ix0[i]) * p.IC + 8 * h) * 4; // bytes (may be negative: only used where the tap is valid)
    }
    const int it0 = (k * p.T) / KS, it1 = ((k + 1) * p.T) / KS, nIt = it1 - it0;
    // weights through a buffer descriptor as well: the lane offset is a constant, the iteration a scalar offset, the four 1 KB pieces immediates --
    // no vector address arithmetic, and no address register for the allocator to recycle as a load destination (which costs a vmcnt(0) per iteration)
    const __amdgpu_buffer_rsrc_t wRsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(wpk), 0, static_cast<int>(p.wBytes), 0x00020000);
    const int wLane = lane * 16;
    const int wTile = ocTile * p.T; // + itp, * 4096 bytes

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;

    // the look-ahead position (scalar): iteration itp = (tap (dyp, dxp), chunk cp); it stops at the wave's last iteration (a few redundant loads at the
    // end instead of a branch around the loads, which would cost the compiler its vmcnt bookkeeping)
    int itp = it0, cp = 0, dxp = 0, dyp = 0;
    if (nIt > 0) {
        const int tap = it0 / p.cpt;
        cp = it0 - tap * p.cpt;
        dyp = tap / p.kw;
        dxp = tap - dyp * p.kw;
    }
    f32x4 A[DEPTH][MI][2], B[DEPTH][2][2];
    auto issue = [&](f32x4 (&a)[MI][2], f32x4 (&b)[2][2]) {
        const int tapOfs = ((dyp * p.W + dxp) * p.IC + 16 * cp) * 4;
        const int wOfs = (wTile + itp) * 4096;
        auto wload = [&](int piece) {
            if (SNNHIP_KS_ABL & 1) return f32x4{1.0f, 2.0f, 3.0f, static_cast<float>(piece)};
            const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wRsrc, wLane + piece * 1024, wOfs, 0);
            f32x4 f;
            __builtin_memcpy(&f, &v, 16);
            return f;
        };
        b[0][0] = wload(0);
        b[0][1] = wload(1);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const bool ok = static_cast<unsigned>(iy0[i] + dyp) < static_cast<unsigned>(p.H) && static_cast<unsigned>(ix0[i] + dxp) < static_cast<unsigned>(p.W);
            const unsigned ofs = ok ? static_cast<unsigned>(base[i] + tapOfs) : kOutOfRange;
            if (SNNHIP_KS_ABL & 2) {
                a[i][0] = f32x4{1.0f, 2.0f, static_cast<float>(ofs), 4.0f};
                a[i][1] = a[i][0];
                continue;
            }
            const i32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(xRsrc, static_cast<int>(ofs), 0, 0);
            const i32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(xRsrc, static_cast<int>(ofs + 16u), 0, 0);
            __builtin_memcpy(&a[i][0], &v0, 16); // (a bit_cast of a vector ELEMENT expression reads element 0 whatever the index: DESIGN.md 5.1)
            __builtin_memcpy(&a[i][1], &v1, 16);
        }
        b[1][0] = wload(2);
        b[1][1] = wload(3);
        if (itp + 1 < it1) { // advance the look-ahead position (scalar unit)
            ++itp;
            if (++cp == p.cpt) {
                cp = 0;
                if (++dxp == p.kw) {
                    dxp = 0;
                    ++dyp;
                }
            }
        }
    };
    auto compute = [&](const f32x4 (&a)[MI][2], const f32x4 (&b)[2][2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q][e], b[j][q][e], acc[i][j], 0, 0, 0);
    };

    // ---- K loop: loads DEPTH - 1 iterations ahead, 16 MI MFMAs per iteration; sched_barrier keeps the next loads in FRONT of this iteration's MFMAs
    // (left alone the scheduler sinks them behind the MFMAs and the wave waits out every load it has just issued)
    if (nIt > 0) {
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) issue(A[d], B[d]);
        for (int i0 = 0; i0 < nIt; i0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (i0 + d < nIt) {
                    issue(A[(d + DEPTH - 1) % DEPTH], B[(d + DEPTH - 1) % DEPTH]);
                    __builtin_amdgcn_sched_barrier(0);
                    compute(A[d], B[d]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    if (SNNHIP_KS_ABL & 4) {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < MI; ++i) t += acc[i][0][0] + acc[i][1][5];
        if (t == 12345.678f) y[0] = t;
        return;
    }
    // ---- reduction over the block's waves through LDS (the two epilogue rows a lane can need are requested first)
    const float4 e4a = epi[ocTile * 64 + r], e4b = epi[ocTile * 64 + 32 + r];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int i = g >> 3, j = (g >> 2) & 1, vq = g & 3;
        red[(k * G + g) * 64 + lane] = f32x4{acc[i][j][4 * vq], acc[i][j][4 * vq + 1], acc[i][j][4 * vq + 2], acc[i][j][4 * vq + 3]};
    }
    __syncthreads();
    const bool whole = (mTile + 1) * (32 * MI) <= p.M; // (block-uniform: only the tensor's last tile can be partial)
    f32x4 sum[GW];
#pragma unroll
    for (int gg = 0; gg < GW; ++gg) {
        const int g = min(k + gg * KS, G - 1); // (a wave without a gg-th quad re-reads the last one and stores nothing)
        sum[gg] = red[g * 64 + lane];
#pragma unroll
        for (int kk = 1; kk < KS; ++kk) sum[gg] += red[(kk * G + g) * 64 + lane]; // fixed order: deterministic
    }
#pragma unroll
    for (int gg = 0; gg < GW; ++gg) {
        const int g = k + gg * KS;
        if (g >= G) break; // (wave-uniform)
        const int i = g >> 3, j = (g >> 2) & 1, vq = g & 3;
        const float4 e4 = j ? e4b : e4a;
        const int m0 = mTile * (32 * MI) + 32 * i + 8 * vq + 4 * h;
        float* const yo = y + static_cast<size_t>(m0) * p.OC + ocTile * 64 + 32 * j + r;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act<SIMPLE>(ac, epi_affine(sum[gg][e], e4, p.useBN), 0.0f);
        if (whole) {
#pragma unroll
            for (int e = 0; e < 4; ++e) yo[static_cast<size_t>(e) * p.OC] = v[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (m0 + e < p.M) yo[static_cast<size_t>(e) * p.OC] = v[e];
        }
    }
}

template <bool SIMPLE, int MI, int KS, int DEPTH>
__global__ __launch_bounds__(64 * KS) void conv2d_ksplit_kernel(KsParams p, ActCfg ac, const float* __restrict__ x, const float4* __restrict__ wpk,
                                                               const float4* __restrict__ epi, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) f32x4 red[];
    ksplit_tile<SIMPLE, MI, KS, DEPTH>(p, ac, x, wpk, epi, y, blockIdx.x, red);
}

// Two problems in ONE launch (snnhip_ctx_group_begin / _end around two independent plans: the 3x3 stride-2 convolution of a ResNet stage entry and the
// 1x1 stride-2 downsample beside it, both reading the same tensor): blocks [0, nA) are tiles of A (64 pixels), the rest tiles of B (32 pixels) -- the
// short ones last, where they fill the CUs the first problem's 1.53 or 3.06 blocks per CU leave idle, instead of a launch of their own behind it.
struct KsPair {
    KsParams a, b;
    ActCfg aca, acb;
    const float *xa, *xb;
    const float4 *wa, *wb, *ea, *eb;
    float *ya, *yb;
    int nA;
};
template <int KS>
__global__ __launch_bounds__(64 * KS) void conv2d_ksplit_pair_kernel(KsPair q) {
    extern __shared__ __attribute__((aligned(16))) f32x4 red[];
    if (static_cast<int>(blockIdx.x) < q.nA) ksplit_tile<true, 2, KS, 2>(q.a, q.aca, q.xa, q.wa, q.ea, q.ya, blockIdx.x, red);
    else ksplit_tile<true, 1, KS, 2>(q.b, q.acb, q.xb, q.wb, q.eb, q.yb, static_cast<int>(blockIdx.x) - q.nA, red);
}

struct KsplitConvPlan;
struct KsDeferred {
    KsplitConvPlan* plan;
    const float* x;
    float* y;
};
struct KsGroup {
    std::vector<KsDeferred> items;
};

struct KsplitConvPlan : ConvPlanBase {
    KsParams p;
    ActCfg ac;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    int mi = 2, ks = 4, depth = 2;
    bool simple = true;
    dim3 grid;
    size_t ldsBytes = 0;

    template <bool S>
    const void* kernel_of() const {
#define KS_PICK(MI_, KS_, D_) \
    if (mi == MI_ && ks == KS_ && depth == D_) return reinterpret_cast<const void*>(conv2d_ksplit_kernel<S, MI_, KS_, D_>);
        KS_PICK(1, 1, 2) KS_PICK(1, 2, 2) KS_PICK(1, 3, 2) KS_PICK(1, 4, 2) KS_PICK(1, 6, 2) KS_PICK(1, 8, 2)
        KS_PICK(2, 1, 2) KS_PICK(2, 2, 2) KS_PICK(2, 3, 2) KS_PICK(2, 4, 2) KS_PICK(2, 6, 2) KS_PICK(2, 8, 2)
        KS_PICK(1, 4, 3) KS_PICK(2, 4, 3)
#undef KS_PICK
        return nullptr;
    }
    const void* kernel() const { return simple ? kernel_of<true>() : kernel_of<false>(); }

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "conv2d: expects 1 input, got %d", nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.H && x->w == p.W && x->c == p.IC, "conv2d: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h, x->w,
                       x->c, p.N, p.H, p.W, p.IC);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.OC, "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n,
                       out->h, out->w, out->c, p.N, p.OH, p.OW, p.OC);
        if (ctx->ksGroup) { // inside snnhip_ctx_group_begin / _end: launched (alone or with its partner) by the group's end
            static_cast<KsGroup*>(ctx->ksGroup)->items.push_back({this, x->data, out->data});
            return SNNHIP_OK;
        }
        return launch(x->data, out->data);
    }
    int launch(const float* xd, float* yd) {
        const float4* w4 = reinterpret_cast<const float4*>(d_w);
        const float4* e4 = reinterpret_cast<const float4*>(d_epi);
        void* args[] = {&p, &ac, &xd, &w4, &e4, &yd};
        if (::snnhip::trace_active()) { // (SNNHIP_LAUNCH for a kernel picked at run time)
            hipEvent_t evS = nullptr, evE = nullptr;
            (void) ::snnhip::trace_events(kernel(), ctx->stream, &evS, &evE);
            SNNHIP_CHECK_HIP(hipExtLaunchKernel(kernel(), grid, dim3(64 * ks), args, ldsBytes, ctx->stream, evS, evE, 0));
        } else {
            SNNHIP_CHECK_HIP(hipLaunchKernel(kernel(), grid, dim3(64 * ks), args, ldsBytes, ctx->stream));
        }
        return SNNHIP_OK;
    }
};

int launch_pair(snnhip_ctx* ctx, const KsDeferred& A, const KsDeferred& B) {
    KsPair q;
    q.a = A.plan->p; q.b = B.plan->p;
    q.aca = A.plan->ac; q.acb = B.plan->ac;
    q.xa = A.x; q.xb = B.x;
    q.wa = reinterpret_cast<const float4*>(A.plan->d_w); q.wb = reinterpret_cast<const float4*>(B.plan->d_w);
    q.ea = reinterpret_cast<const float4*>(A.plan->d_epi); q.eb = reinterpret_cast<const float4*>(B.plan->d_epi);
    q.ya = A.y; q.yb = B.y;
    q.nA = static_cast<int>(A.plan->grid.x);
    const int ks = A.plan->ks;
    const void* fn = ks == 2 ? reinterpret_cast<const void*>(conv2d_ksplit_pair_kernel<2>) : ks == 4 ? reinterpret_cast<const void*>(conv2d_ksplit_pair_kernel<4>)
                                                                                                 : reinterpret_cast<const void*>(conv2d_ksplit_pair_kernel<8>);
    const size_t lds = static_cast<size_t>(ks) * 16 * 64 * sizeof(float4); // A's (two pixel tiles per wave) covers B's
    static bool ldsSet[3] = {false, false, false}; // (once per kernel: not inside every recorded launch)
    bool& set = ldsSet[ks == 2 ? 0 : ks == 4 ? 1 : 2];
    if (lds > 64 * 1024 && !set) {
        SNNHIP_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        set = true;
    }
    const dim3 grid(A.plan->grid.x + B.plan->grid.x);
    void* args[] = {&q};
    const double sharedIn = A.x == B.x ? 4.0 * A.plan->p.N * A.plan->p.H * A.plan->p.W * A.plan->p.IC : 0.0; // the tensor both read: once
    TraceScope ts("conv2d_ksplit pair [" + A.plan->desc + "] + [" + B.plan->desc + "]", A.plan->flops + B.plan->flops, A.plan->bytes + B.plan->bytes - sharedIn);
    if (::snnhip::trace_active()) {
        hipEvent_t evS = nullptr, evE = nullptr;
        (void) ::snnhip::trace_events(fn, ctx->stream, &evS, &evE);
        SNNHIP_CHECK_HIP(hipExtLaunchKernel(fn, grid, dim3(64 * ks), args, lds, ctx->stream, evS, evE, 0));
    } else {
        SNNHIP_CHECK_HIP(hipLaunchKernel(fn, grid, dim3(64 * ks), args, lds, ctx->stream));
    }
    return SNNHIP_OK;
}

} // namespace

// ---- launch groups (include/snnhip.h: snnhip_ctx_group_begin / _end): K-split plans run between the two calls are launched by the group's end -- two
// that fit one grid (a 64-pixel-tile problem split over 2 / 4 / 8 waves and a 32-pixel-tile one, simple activations) as ONE launch, anything else one by one
bool ksplit_plan(const snnhip_plan* plan) { return dynamic_cast<const KsplitConvPlan*>(plan) != nullptr; }
int ksplit_group_begin(snnhip_ctx* ctx) {
    SNNHIP_REQUIRE(!ctx->ksGroup, "ctx_group_begin: a group is already open on this context");
    ctx->ksGroup = new KsGroup();
    return SNNHIP_OK;
}
int ksplit_group_end(snnhip_ctx* ctx) {
    SNNHIP_REQUIRE(ctx->ksGroup, "ctx_group_end: no group is open on this context");
    KsGroup* g = static_cast<KsGroup*>(ctx->ksGroup);
    ctx->ksGroup = nullptr;
    int rc = SNNHIP_OK;
    auto fits = [](const KsDeferred& a, const KsDeferred& b) {
        return a.plan->mi == 2 && b.plan->mi == 1 && a.plan->simple && b.plan->simple && a.plan->depth == 2 && (a.plan->ks == 2 || a.plan->ks == 4 || a.plan->ks == 8) &&
               a.plan->ks <= 8 && !SNNHIP_KS_ABL;
    };
    if (g->items.size() == 2 && !snnhip::option("SNNHIP_KSPLIT_NO_PAIRS") && (fits(g->items[0], g->items[1]) || fits(g->items[1], g->items[0]))) {
        const bool first = fits(g->items[0], g->items[1]);
        rc = launch_pair(ctx, g->items[first ? 0 : 1], g->items[first ? 1 : 0]);
    } else {
        for (const KsDeferred& d : g->items) {
            TraceScope ts(d.plan);
            const int r1 = d.plan->launch(d.x, d.y);
            if (r1 != SNNHIP_OK) rc = r1;
        }
    }
    delete g;
    return rc;
}

int make_conv2d_ksplit_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    // eligibility: fp32, zero padding by clipped taps, no fused neighbours, whole 16-channel iterations and 64-channel output tiles
    if (g.dtype != SNNHIP_F32 || g.preMode != 0 || g.preShift != 0 || g.normShift || g.addAct >= 0) return SNNHIP_E_UNSUPPORTED;
    if (g.padMode != SNNHIP_PAD_CONSTANT && g.padMode != SNNHIP_PAD_NONE) return SNNHIP_E_UNSUPPORTED;
    if (g.IC % 16 != 0 || g.OC % 64 != 0 || g.act == SNNHIP_ACT_SILU_QUIRK) return SNNHIP_E_UNSUPPORTED;
    if (g.sh < 1 || g.sh > 2 || g.sw < 1 || g.sw > 2 || g.kh > 7 || g.kw > 7) return SNNHIP_E_UNSUPPORTED;
    const double inBytes = 4.0 * g.N * g.H * g.W * g.IC, outCount = static_cast<double>(g.N) * g.OH * g.OW * g.OC;
    if (inBytes >= 2147483647.0 || outCount >= 2147483647.0) return SNNHIP_E_UNSUPPORTED; // 31-bit byte offsets into the input (kOutOfRange), 32-bit pixel index
    const char* force = snnhip::option("SNNHIP_CONV");
    const bool forced = force && strcmp(force, "ksplit") == 0;
    // default: the 3x3 stride-2 layers (the shapes conv2d_mfma_kernel runs split-K + a reduce launch on) and their 1x1 stride-2 siblings (ResNet's
    // downsample branch: few short waves, where conv1x1_stream_kernel's weight staging + barrier in front of the first MFMA is a third of a wave's life:
    // 34.9 -> 27.0 us for the three layers at batch 32, tools/r6_ds2.sh); anything else only when forced
    const bool s2 = g.sh == 2 && g.sw == 2 && g.IC >= 32;
    if (!forced && !(s2 && ((g.kh == 3 && g.kw == 3) || (g.kh == 1 && g.kw == 1)))) return SNNHIP_E_UNSUPPORTED;

    KsParams p = {};
    p.N = g.N; p.H = g.H; p.W = g.W; p.IC = g.IC; p.OC = g.OC; p.OH = g.OH; p.OW = g.OW;
    p.kh = g.kh; p.kw = g.kw; p.sh = g.sh; p.sw = g.sw; p.padx = g.padx; p.pady = g.pady;
    p.M = g.N * g.OH * g.OW;
    p.cpt = g.IC / 16;
    p.T = g.kh * g.kw * p.cpt;
    p.useBN = g.useBN;
    p.xBytes = static_cast<unsigned>(inBytes);
    p.wBytes = static_cast<unsigned>(static_cast<size_t>(g.OC / 64) * p.T * 4096);

    // (MI, KS), fitted on ResNet-18's three stage entries at batch 32 (tools/r6_ks.sh): a wave should run about 18 K iterations -- fewer
    // and its set-up, first loads and reduction weigh too much, more and the launch has too few waves to keep every SIMD busy to the end --, and a
    // launch should have at least 1.5 waves per SIMD: 32-pixel tiles when 64-pixel tiles give fewer.
    const int simds = (ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256) * 4;
    int mi = 2, ks = 1, depth = 2;
    if (g.kh * g.kw == 1) { // pointwise: four iterations per wave (4 / 8 / 16 iterations -> 1 / 2 / 4 waves), 32-pixel tiles
        mi = 1;
        while (ks < 8 && p.T / (2 * ks) >= 4) ks *= 2;
    } else {
        while (ks < 8 && p.T / (2 * ks) >= 14) ks *= 2; // 36 iterations -> 2 waves, 72 -> 4, 144 -> 8
        if (static_cast<double>(up_div(p.M, 64)) * (g.OC / 64) * ks < 1.5 * simds) mi = 1;
    }
    if (const char* e = snnhip::option("SNNHIP_KSPLIT")) { // experiments: MI,KS[,DEPTH]
        int a = 0, b = 0, c = 0;
        const int n = sscanf(e, "%d,%d,%d", &a, &b, &c);
        if (n >= 2) {
            mi = a;
            ks = b;
        }
        if (n >= 3) depth = c;
    }

    auto* plan = new KsplitConvPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * g.kh * g.kw);
    plan->epi4 = epi4;
    plan->p = p;
    plan->mi = mi;
    plan->ks = ks;
    plan->depth = depth;
    plan->p.mTiles = up_div(p.M, 32 * mi);
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->simple = act_is_simple(g.act);
    plan->dtype = SNNHIP_F32;
    plan->ldsBytes = static_cast<size_t>(ks) * 8 * mi * 64 * sizeof(float4);
    plan->grid = dim3(static_cast<unsigned>(plan->p.mTiles) * (g.OC / 64));
    if (!plan->kernel() || plan->ldsBytes > 160 * 1024) {
        set_error("conv2d_ksplit: no instantiation for MI=%d KS=%d DEPTH=%d", mi, ks, depth);
        delete plan;
        return SNNHIP_E_UNSUPPORTED;
    }
    if (plan->ldsBytes > 64 * 1024 &&
        hipFuncSetAttribute(plan->kernel(), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(plan->ldsBytes)) != hipSuccess) {
        set_error("conv2d_ksplit: hipFuncSetAttribute(%zu) failed", plan->ldsBytes);
        delete plan;
        return SNNHIP_E_HIP;
    }

    // weights: [oc tile][iteration = (tap, 16-channel chunk)][n tile j][half q][lane = 32 h + col] float4 over e: channel 16 c + 8 h + 4 q + e
    std::vector<float> W(static_cast<size_t>(g.OC / 64) * p.T * 1024, 0.0f);
    for (int ot = 0; ot < g.OC / 64; ++ot)
        for (int it = 0; it < p.T; ++it) {
            const int tap = it / p.cpt, c = it % p.cpt, dy = tap / g.kw, dx = tap % g.kw;
            for (int j = 0; j < 2; ++j)
                for (int q = 0; q < 2; ++q)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 4; ++e) {
                            const int oc = ot * 64 + 32 * j + (l & 31), ch = 16 * c + 8 * (l >> 5) + 4 * q + e;
                            W[((static_cast<size_t>(ot) * p.T + it) * 4 + 2 * j + q) * 256 + l * 4 + e] =
                                w_oihw[((static_cast<size_t>(oc) * g.IC + ch) * g.kh + dy) * g.kw + dx];
                        }
        }
    int rc = plan->upload(W.data(), W.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epi4.data(), epi4.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = g.N; plan->inDims[1] = g.H; plan->inDims[2] = g.W; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->flops = 2.0 * g.kh * g.kw * g.IC * g.OC * static_cast<double>(g.OH) * g.OW * g.N;
    plan->bytes = 4.0 * (static_cast<double>(g.N) * g.H * g.W * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC + static_cast<double>(g.OC) * g.IC * g.kh * g.kw);
    char buf[320];
    snprintf(buf, sizeof(buf), "conv2d_mfma_f32_32x32x2 k=%dx%d s=%d ic=%d oc=%d ksplit: tile=%dpx x 64oc, K over %d waves (%d iterations of 16 channels), operands from L2, depth %d, lds=%zuB, grid %u",
             g.kh, g.kw, g.sh, g.IC, g.OC, 32 * mi, ks, p.T, depth, plan->ldsBytes, plan->grid.x);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
