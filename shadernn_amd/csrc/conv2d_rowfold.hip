// conv2d_rowfold.hip -- fp16 k x k stride-1 convolution with FEW output channels (k * OC <= 32: the image-producing 9x9 32->3 layer of the
// fast-neural-style networks, BASELINE configs[4]) on v_mfma_f32_32x32x16_f16, with the kernel's COLUMNS folded into the GEMM's N.
//
// With 3 output channels a 32-wide MFMA column block is 29/32 padding, which is why conv2d_thin.hip runs these layers on the 4x4x4 MFMA (at
// 113 TFLOP/s: 1.26 ms of Candy's 8.7 ms batch-8 pass).  But the sum over the kernel column fx commutes with everything else:
//     out[y][x][oc] = sum_fx  P[y][x + fx][fx][oc],        P[y][x'][fx][oc] = sum_{fy, ic} W[oc][ic][fy][fx] * in[y + fy - pady][x' - padx'][ic]
// P is a GEMM with M = pixels x' of a row, N = (fx, oc) = 27 of 32 columns useful, K = (fy, ic) = 9 x 32 = 288 -- a dense 32x32x16 problem -- and
// the outer sum is a 9-term shift-add over neighbouring pixels, done from LDS.  Same operator contract as the other convolution kernels
// (shadertemplate_vk_conv2d.comp:148-347: padding modes, bias -> BN -> activation), the Pad layer in front can be fused into the staging (rule D).
//
// Block = 256 threads = 4 waves, output tile 8 rows x (64 - k + 1) columns of one image:
//   input tile (8 + k - 1 rows x 64 columns x IC halfs) staged once in LDS, 16-byte slots XOR-swizzled by the column so that the ds_read_b128 of
//   an A operand (lane = (column, K half)) is conflict-free; all k * IC/16 B operands (weights) of a lane live in REGISTERS for the whole block
//   wave = 2 output rows x both 32-column MFMA tiles: loop over the input rows r it needs, per row and 16-channel step two A reads, and for each
//   of its output rows j with a valid tap fy = r - j two MFMAs (64 accumulators); 72 MFMAs per wave for k = 9, IC = 32
//   epilogue: the four waves' P tiles go to LDS (over the input tile), every thread shift-adds its output pixels, epilogue, 2-byte stores.
#include <cstring>
#include <vector>

#include "epilogue.h"
#include "snnhip_internal.h"

namespace snnhip {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct RowfoldParams {
    int N, H, W, IC, OC, OH, OW, padx, pady, padMode, useBN;
    int preMode, preX, preY, srcH, srcW, preShift; // fused Pad / nearest x2 upsampling in front (ConvGeom)
    int tilesX, tilesY;
    // InstanceNorm in front (graph rule I): normAc(x * mul[n][c] + shift[n][c]) applied to the staged values; null = none
    const float* normShift;
    const float* normMul;
    ActCfg normAc;
};

constexpr int kTH = 8, kCols = 64;
constexpr int kPP = 33; // floats per pixel of the P tile: 32 columns + 1, so that the shift-add's reads (consecutive lanes = consecutive pixels) spread over the banks

template <int K, int ICS /* IC / 16 */, bool SIMPLE>
__global__ __launch_bounds__(256, 2) void conv2d_rowfold_kernel(RowfoldParams p, ActCfg ac, const _Float16* __restrict__ x, const float4* __restrict__ wp,
                                                               const float4* __restrict__ epi, _Float16* __restrict__ y) {
    constexpr int ROWS = kTH + K - 1;     // staged input rows
    constexpr int TW = kCols - K + 1;     // output columns per tile
    constexpr int Q = 2 * ICS;            // 16-byte slots per pixel
    constexpr int NK = K * ICS;           // K steps (weights kept in registers)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int mt = blockIdx.x;
    const int tx = mt % p.tilesX, ty = (mt / p.tilesX) % p.tilesY, n = mt / (p.tilesX * p.tilesY);
    const int ox0 = tx * TW, oy0 = ty * kTH;
    const int ix0 = ox0 - p.padx, iy0 = oy0 - p.pady;

    // ---- this lane's B operands: step s = fy * ICS + c -> 8 halfs {W[oc][16c + 8h + j][fy][fx]} for column n = fx * OC + oc = l32
    float4 b[NK];
#pragma unroll
    for (int s = 0; s < NK; ++s) b[s] = wp[s * 64 + lane];

    // ---- stage the input tile; zero outside the (padded) image, the Pad layer resolved on the fly.  Element e = tid + 256 r of the tile
    // [ROWS][64 columns][Q slots] is column (tid / Q) % 64, slot tid % Q of row r * RPI + tid / (64 Q): a thread's elements share their column and
    // channel slot, and a round's row is wave-uniform -- the column is resolved once per thread, the row on the scalar unit, an element costs an
    // address add, its load, and the LDS store at an immediate offset.  (Resolving row AND column per element, with a bounds branch each, made this
    // prologue 1300 VALU + 1300 SALU instructions for the wave's 72 MFMAs: the kernel was issue-bound on address arithmetic.)
    {
        constexpr int CPR = kCols * Q;     // elements per staged row (256 or 128)
        constexpr int RPI = 256 / CPR;     // staged rows per round of 256 elements (1 or 2)
        constexpr int NRND = (ROWS + RPI - 1) / RPI;
        const _Float16* xn = x + static_cast<size_t>(n) * p.srcH * p.srcW * p.IC;
        const int sl = tid % Q, c = (tid / Q) % kCols;
        const int rsub = __builtin_amdgcn_readfirstlane(tid / CPR); // wave-uniform (a wave's 64 elements sit in one row)
        int sx = resolve_nobranch(ix0 + c, p.W, p.padMode);
        if (p.preMode) { // (uniform) a pixel of the padded image -> the source pixel the Pad layer would have copied; -1 stays -1
            const int px = resolve_nobranch(sx - p.preX, p.srcW << p.preShift, p.preMode);
            sx = sx < 0 ? -1 : (px < 0 ? -1 : px >> p.preShift);
        }
        const bool colOk = sx >= 0;
        const int colOfs = (colOk ? sx : 0) * p.IC + 8 * sl;
        float* const ldsp = smem + ((rsub * kCols + c) * Q + (sl ^ ((c >> (Q == 4 ? 2 : 3)) & (Q - 1)))) * 4;
        // graph rule I: a thread's elements share their 8 channels and, within a block, the image
        float nShift[8], nMul[8];
        if (p.normShift) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                nShift[k] = p.normShift[static_cast<size_t>(n) * p.IC + 8 * sl + k];
                nMul[k] = p.normMul[static_cast<size_t>(n) * p.IC + 8 * sl + k];
            }
        }
        const bool nRelu = p.normAc.act == SNNHIP_ACT_RELU; // (uniform) one instruction instead of the general mul / max / med3
        float4 v[NRND];
        bool rowOk[NRND];
        // all of a thread's loads (16 for a 16 x 64 x 32-channel tile) are requested before the first LDS store: one HBM round trip per block
#pragma unroll
        for (int r = 0; r < NRND; ++r) {
            const int rr = r * RPI + rsub;
            int sy = resolve_nobranch(iy0 + rr, p.H, p.padMode);
            if (p.preMode) {
                const int py = resolve_nobranch(sy - p.preY, p.srcH << p.preShift, p.preMode);
                sy = sy < 0 ? -1 : (py < 0 ? -1 : py >> p.preShift);
            }
            rowOk[r] = sy >= 0 && rr < ROWS; // (uniform)
            v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rowOk[r]) v[r] = *reinterpret_cast<const float4*>(xn + static_cast<size_t>(sy) * p.srcW * p.IC + colOfs); // (a column outside reads column 0: masked below)
        }
        if (p.normShift) { // (the activation test sits OUTSIDE the element loops: inside, the compiler kept both forms behind a branch per element)
            if (nRelu) {
#pragma unroll
                for (int r = 0; r < NRND; ++r) {
                    h8 hv = *reinterpret_cast<const h8*>(&v[r]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) hv[k] = static_cast<_Float16>(fmaxf(fmaf(static_cast<float>(hv[k]), nMul[k], nShift[k]), 0.0f));
                    v[r] = *reinterpret_cast<const float4*>(&hv);
                }
            } else {
#pragma unroll
                for (int r = 0; r < NRND; ++r) {
                    h8 hv = *reinterpret_cast<const h8*>(&v[r]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float f = fmaf(static_cast<float>(hv[k]), nMul[k], nShift[k]);
                        hv[k] = static_cast<_Float16>(__builtin_amdgcn_fmed3f(fmaxf(f, f * p.normAc.alpha), p.normAc.lo, p.normAc.hi));
                    }
                    v[r] = *reinterpret_cast<const float4*>(&hv);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NRND; ++r) {
            const bool live = rowOk[r] && colOk; // padding stays zero (the norm is not applied to it)
            const float4 o = make_float4(live ? v[r].x : 0.f, live ? v[r].y : 0.f, live ? v[r].z : 0.f, live ? v[r].w : 0.f);
            if (r * RPI + rsub < ROWS) *reinterpret_cast<float4*>(ldsp + r * RPI * CPR * 4) = o;
        }
    }
    __syncthreads();

    // ---- wave = output rows 2w, 2w + 1 of the tile x both column tiles
    f32x16 acc[2][2]; // [output row j][column tile]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][t][i] = 0.0f;
    const int row0 = 2 * wave;
#pragma unroll
    for (int rr = 0; rr < K + 1; ++rr) { // input row row0 + rr feeds output row j at tap fy = rr - j
#pragma unroll
        for (int c = 0; c < ICS; ++c) {
            float4 a[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int col = t * 32 + l32, s = 2 * c + h;
                a[t] = *reinterpret_cast<const float4*>(smem + (((row0 + rr) * kCols + col) * Q + (s ^ ((col >> (Q == 4 ? 2 : 3)) & (Q - 1)))) * 4);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int fy = rr - j;
                if (fy < 0 || fy >= K) continue; // compile-time after unrolling
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&a[t]), *reinterpret_cast<const h8*>(&b[fy * ICS + c]), acc[j][t], 0, 0, 0);
            }
        }
    }
    __syncthreads(); // every wave is done with the input tile: its LDS becomes the P tiles [8 rows][64 columns][33] floats

    // D layout: lane holds column l32 and pixel rows 8 (i / 4) + 4 h + i % 4 of each 32-pixel tile
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int px = t * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                smem[((row0 + j) * kCols + px) * kPP + l32] = acc[j][t][i];
            }
    __syncthreads();

    // ---- shift-add + epilogue: thread -> output pixels (row, x), all OC channels
    for (int o = tid; o < kTH * TW; o += 256) {
        const int r = o / TW, xx = o - r * TW;
        const int oy = oy0 + r, ox = ox0 + xx;
        if (oy >= p.OH || ox >= p.OW) continue;
        const float* pr = smem + (r * kCols + xx) * kPP;
        _Float16* yp = y + ((static_cast<size_t>(n) * p.OH + oy) * p.OW + ox) * p.OC;
        for (int oc = 0; oc < p.OC; ++oc) {
            float s = 0.0f;
#pragma unroll
            for (int fx = 0; fx < K; ++fx) s += pr[fx * kPP + fx * p.OC + oc]; // P[x + fx][fx][oc]
            float v = epi_affine(s, epi[oc], p.useBN);
            v = SIMPLE ? apply_act<true>(ac, v, 0.0f) : epi_act(ac.act, ac.leaky, v, 0.0f);
            yp[oc] = static_cast<_Float16>(v);
        }
    }
}

struct RowfoldPlan : ConvPlanBase {
    RowfoldParams p;
    ActCfg ac;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    size_t ldsBytes = 0;
    dim3 grid;
    void (*kernel)(RowfoldParams, ActCfg, const _Float16*, const float4*, const float4*, _Float16*) = nullptr;

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "conv2d: expects 1 input, got %d", nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == p.N && x->h == p.srcH && x->w == p.srcW && x->c == p.IC, "conv2d: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h,
                       x->w, x->c, p.N, p.srcH, p.srcW, p.IC);
        SNNHIP_REQUIRE(out->n == p.N && out->h == p.OH && out->w == p.OW && out->c == p.OC, "conv2d: output dims %dx%dx%dx%d != plan %dx%dx%dx%d", out->n,
                       out->h, out->w, out->c, p.N, p.OH, p.OW, p.OC);
        SNNHIP_LAUNCH(kernel, grid, dim3(256), ldsBytes, ctx->stream, p, ac, reinterpret_cast<const _Float16*>(x->data), reinterpret_cast<const float4*>(d_w),
                           reinterpret_cast<const float4*>(d_epi), reinterpret_cast<_Float16*>(out->data));
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

typedef void (*RowfoldFn)(RowfoldParams, ActCfg, const _Float16*, const float4*, const float4*, _Float16*);
template <int K>
RowfoldFn pick_rowfold(int ics, bool simple) {
    if (ics == 1) return simple ? conv2d_rowfold_kernel<K, 1, true> : conv2d_rowfold_kernel<K, 1, false>;
    return simple ? conv2d_rowfold_kernel<K, 2, true> : conv2d_rowfold_kernel<K, 2, false>;
}

} // namespace

int make_conv2d_rowfold_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out) {
    {   // the row-marching form first (conv2d_rowmarch.hip: OC <= 4); this file's tile kernel takes what it declines (and SNNHIP_ROWFOLD=tile)
        const char* force = snnhip::option("SNNHIP_CONV");
        if (!force || strcmp(force, "rowfold") == 0) {
            const int rc = make_conv2d_rowmarch_plan(ctx, g, w_oihw, epi4, out);
            if (rc != SNNHIP_E_UNSUPPORTED) return rc;
        }
    }
    // eligibility: half tensors, square odd kernel 5 / 7 / 9, stride 1, k * OC <= 32, IC = 16 or 32 (the weights of a lane stay in registers)
    const char* force = snnhip::option("SNNHIP_CONV");
    if (force && strcmp(force, "rowfold") != 0) return SNNHIP_E_UNSUPPORTED;
    if (g.normShift && !act_is_simple(g.normAct)) return SNNHIP_E_UNSUPPORTED;
    if (g.dtype != SNNHIP_F16 || g.kh != g.kw || (g.kh != 5 && g.kh != 7 && g.kh != 9) || g.sh != 1 || g.sw != 1) return SNNHIP_E_UNSUPPORTED;
    if (g.kh * g.OC > 32 || (g.IC != 16 && g.IC != 32) || g.act == SNNHIP_ACT_SILU_QUIRK || g.addAct >= 0) return SNNHIP_E_UNSUPPORTED;
    if (static_cast<double>(g.N) * g.H * g.W * g.IC >= 2147483647.0) return SNNHIP_E_UNSUPPORTED;
    const int K = g.kh, ICS = g.IC / 16, TW = kCols - K + 1;
    RowfoldParams p = {};
    p.N = g.N; p.H = g.H; p.W = g.W; p.IC = g.IC; p.OC = g.OC; p.OH = g.OH; p.OW = g.OW; p.padx = g.padx; p.pady = g.pady; p.padMode = g.padMode; p.useBN = g.useBN;
    p.preMode = g.preMode; p.preX = g.preX; p.preY = g.preY; p.preShift = g.preShift;
    p.srcH = g.preMode ? g.srcH : g.H;
    p.srcW = g.preMode ? g.srcW : g.W;
    p.tilesX = up_div(g.OW, TW);
    p.tilesY = up_div(g.OH, kTH);
    p.normShift = g.normShift; p.normMul = g.normMul;
    p.normAc = make_act_cfg(g.normShift ? g.normAct : SNNHIP_ACT_NONE, g.normLeaky);
    const size_t lds = std::max(static_cast<size_t>(kTH + K - 1) * kCols * g.IC * 2, static_cast<size_t>(kTH) * kCols * kPP * sizeof(float));
    const bool simple = act_is_simple(g.act);
    RowfoldFn fn = K == 9 ? pick_rowfold<9>(ICS, simple) : K == 7 ? pick_rowfold<7>(ICS, simple) : pick_rowfold<5>(ICS, simple);
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
        set_error("conv2d_rowfold: hipFuncSetAttribute(%zu) failed", lds);
        return SNNHIP_E_HIP;
    }
    auto* plan = new RowfoldPlan();
    plan->ctx = ctx;
    plan->g = g;
    plan->w_oihw.assign(w_oihw, w_oihw + static_cast<size_t>(g.OC) * g.IC * K * K);
    plan->epi4 = epi4;
    plan->p = p;
    plan->ac = make_act_cfg(g.act, g.leaky);
    plan->kernel = fn;
    plan->ldsBytes = lds;
    plan->grid = dim3(p.tilesX * p.tilesY * g.N);
    plan->dtype = SNNHIP_F16;
    // weights: Wp[step = fy * ICS + c][lane = 32 h + n] x 8 halfs {W[oc][16 c + 8 h + j][fy][fx]}, n = fx * OC + oc; columns >= K * OC are zero
    const int NK = K * ICS;
    std::vector<float> wpk(static_cast<size_t>(NK) * 64 * 4, 0.0f);
    _Float16* wph = reinterpret_cast<_Float16*>(wpk.data());
    for (int oc = 0; oc < g.OC; ++oc)
        for (int ic = 0; ic < g.IC; ++ic)
            for (int fy = 0; fy < K; ++fy)
                for (int fx = 0; fx < K; ++fx) {
                    const int c = ic / 16, hh = (ic % 16) / 8, j = ic % 8, nn = fx * g.OC + oc;
                    wph[((static_cast<size_t>(fy) * ICS + c) * 64 + hh * 32 + nn) * 8 + j] =
                        static_cast<_Float16>(w_oihw[((static_cast<size_t>(oc) * g.IC + ic) * K + fy) * K + fx]);
                }
    int rc = plan->upload(wpk.data(), wpk.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epi4.data(), epi4.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = g.N; plan->inDims[1] = p.srcH; plan->inDims[2] = p.srcW; plan->inDims[3] = g.IC;
    plan->outDims[0] = g.N; plan->outDims[1] = g.OH; plan->outDims[2] = g.OW; plan->outDims[3] = g.OC;
    plan->flops = 2.0 * K * K * g.IC * g.OC * static_cast<double>(g.OH) * g.OW * g.N;
    plan->bytes = 2.0 * (static_cast<double>(g.N) * p.srcH * p.srcW * g.IC + static_cast<double>(g.N) * g.OH * g.OW * g.OC + static_cast<double>(g.OC) * g.IC * K * K);
    char buf[256];
    snprintf(buf, sizeof(buf), "conv2d_rowfold_mfma_f16_32x32x16 k=%dx%d s=1 ic=%d oc=%d (columns folded into N: %d of 32) tile=%dx%dpx lds=%zuB", K, K, g.IC, g.OC, K * g.OC,
             kTH, TW, lds);
    plan->desc = buf;
    if (g.preMode) plan->desc += " +pad(" + std::string(g.preMode == SNNHIP_PAD_REFLECT ? "reflect" : g.preMode == SNNHIP_PAD_REPLICATE ? "replicate" : "constant") + ")";
    if (g.preMode && g.preShift) plan->desc += " +upsample(x2)";
    if (g.normShift) plan->desc = "instancenorm(act=" + std::to_string(g.normAct) + ", in the staging) -> " + plan->desc;
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
