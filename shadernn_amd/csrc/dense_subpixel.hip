// dense_subpixel.hip -- fully connected layer and the ESPCN depth-to-space tail.
//
// dense: replaces the reference's CPU/Eigen path DenseLayer::computeImageTexture -> CPUCommonUtil::transform
//   (core/src/ic2/denselayer.cpp:27-38, cpulayer.h:136-171) and its GPU twin shadertemplate_vk_dense.comp:53-79.
//   y[b][o] = act(sum_i W[o*In+i] * x[b][i] + bias[o]).  One wave64 per output row: lanes stride over In with 16-byte
//   loads, wave-level xor-shuffle reduction, lane 0 applies the epilogue (the reference's shader is 1 thread/output).
//   Activation semantics follow the CPU path (cpulayer.h:185-261), including softmax over the output row.
// subpixel: replaces shadertemplate_vk_subpixel.comp:43-71 (depth-to-space(2) + tanh), both the true d2s channel
//   selection (fs_subpixel.glsl:41-64) and the Vulkan shader's depth-slice quirk.
#include "epilogue.h"
#include "plan_util.h"
#include "snnhip_internal.h"

namespace snnhip {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// TX: element type of the activations (float or half); weights, arithmetic and the output row stay fp32 -- a half output tensor is written
// by convert_rows_kernel after the (optional) softmax so that no intermediate is rounded
// A wave owns one output unit for kBatchPerWave batch rows: the weight row is loaded once per group instead of once per batch row (at batch
// 32 the 1280 x 1000 classifier pulled its 5 MB of weights through L2 32 times); per (row, unit) the lane partial sums are formed in the
// same order as before, so results are bit-identical.
// (GB = 4, or 16 from batch 64 on: the 256-image classifier of BASELINE config 4 read its weights 64 times with 4.)
template <bool VEC, typename TX, int kBatchPerWave>
__global__ __launch_bounds__(256) void dense_kernel(int In, int Out, int batch, int act, float leaky, const TX* __restrict__ x,
                                                    const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b0 = blockIdx.y * kBatchPerWave;
    if (o >= Out) return;
    const float* wr = w + static_cast<size_t>(o) * In;
    float acc[kBatchPerWave];
#pragma unroll
    for (int g = 0; g < kBatchPerWave; ++g) acc[g] = 0.0f;
    if (VEC) {
        const float4* w4 = reinterpret_cast<const float4*>(wr);
        for (int i = lane; i < In / 4; i += 64) {
            const float4 a = w4[i];
#pragma unroll
            for (int g = 0; g < kBatchPerWave; ++g) {
                if (b0 + g < batch) { // wave-uniform
                    float v[4];
                    ldv<TX, 4>(x + static_cast<size_t>(b0 + g) * In + 4 * i, v);
                    acc[g] = fmaf(a.x, v[0], acc[g]);
                    acc[g] = fmaf(a.y, v[1], acc[g]);
                    acc[g] = fmaf(a.z, v[2], acc[g]);
                    acc[g] = fmaf(a.w, v[3], acc[g]);
                }
            }
        }
    } else {
        for (int i = lane; i < In; i += 64) {
            const float a = wr[i];
#pragma unroll
            for (int g = 0; g < kBatchPerWave; ++g)
                if (b0 + g < batch) acc[g] = fmaf(a, static_cast<float>(x[static_cast<size_t>(b0 + g) * In + i]), acc[g]);
        }
    }
#pragma unroll
    for (int g = 0; g < kBatchPerWave; ++g) {
        if (b0 + g >= batch) break;
        const float s = wave_sum(acc[g]);
        if (lane == 0) {
            float v = s + bias[o];
            switch (act) {
            case SNNHIP_DENSE_RELU: v = v > 0 ? v : 0.0f * v; break;            // leakyRelu(val, 0.0) cpulayer.h:187,204
            case SNNHIP_DENSE_LEAKY: v = v > 0 ? v : leaky * v; break;
            case SNNHIP_DENSE_SIGMOID: v = 1.0f / (1.0f + expf(-v)); break;
            case SNNHIP_DENSE_TANH: v = (expf(2 * v) - 1) / (expf(2 * v) + 1); break; // cpulayer.h:195
            default: break; // identity, SiLU (no-op in the reference), softmax (second kernel)
            }
            y[static_cast<size_t>(b0 + g) * Out + o] = v;
        }
    }
}

// Batched classifier heads (batch >= 32, fp32, In % 8 == 0: MobileNetV2's 1280 -> 1000 at batch 256, BASELINE configs[3]) are a small GEMM
// y[b][o] = sum_i x[b][i] W[o][i], and dense_kernel above -- one wave per output unit, the batch rows re-read per group -- took 85 us for what is 0.66
// GFLOP and 6.3 MB of operands.  Here: v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain) with the weight rows as the A operand (M = 32 output units) and
// the batch rows as B (N = 32 images): both operands are row-major with K contiguous, so lane (r = l % 32, kh = l / 32) loads the 16 bytes
// [r][8 j + 4 kh ..] of each and the four MFMAs of a K block consume the four components (the K permutation is the same on both sides).  A block =
// one 32 x 32 tile, its four waves split K and add their tiles up through LDS; bias and the CPU path's activation in the epilogue.  (That first form is
// described here for the record; what runs is the LDS-staged form below.)
typedef float f32x16d __attribute__((ext_vector_type(16)));
// Operands through LDS (round 3, second form).  Loaded straight into the operand layout a lane's 16 bytes are 1/8 of a 128-byte line and the other
// seven eighths belong to three more instructions and the other lane half: the four-wave form took 41 us (MobileNetV2) / 19.8 us (ResNet-18) at 0.09 /
// 0.008 of the matrix pipe with 16 MB of HBM traffic, and neither more waves nor more loads in flight moved it -- the lines were
// fetched from L2 over and over.  Here a K chunk of 64 of both 32-row tiles is copied with whole-line loads (16 lanes x 16 bytes = one 256-byte row
// segment; chunk c + 1 is in registers while chunk c is multiplied), rows 68 floats apart in LDS (a multiple of 16 bytes that is odd in 16-byte units:
// the ds_read_b128 of 32 rows at one K offset is conflict-free), and wave w multiplies K sixteenth-pairs [16w, 16w + 16) of every chunk.
constexpr int kDenseKC = 64, kDensePitch = kDenseKC + 4;
__global__ __launch_bounds__(256) void dense_mfma_kernel(int In, int Out, int batch, int act, float leaky, const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float tileW[32 * kDensePitch], tileX[32 * kDensePitch];
    __shared__ float red[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, kh = lane >> 5;
    const int o0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    // staging: thread t copies float4 (row t / 16, column 4 (t % 16)) and the same column of row 16 + t / 16, of both tiles
    const int srow = tid >> 4, scol = (tid & 15) * 4;
    // row starts WITHOUT the column: a thread whose column lies past the row (In < 64, or the last chunk) falls back to column 0 of its row, an
    // address inside the allocation whatever In is (with the column in the pointer the fallback read up to 220 bytes past the last row)
    const float* wsrc0 = w + static_cast<size_t>(min(o0 + srow, Out - 1)) * In;
    const float* wsrc1 = w + static_cast<size_t>(min(o0 + srow + 16, Out - 1)) * In;
    const float* xsrc0 = x + static_cast<size_t>(min(b0 + srow, batch - 1)) * In;
    const float* xsrc1 = x + static_cast<size_t>(min(b0 + srow + 16, batch - 1)) * In;
    float4 pw0, pw1, px0, px1;
    // (In % 8 == 0, scol % 4 == 0: a float4 is inside the row or past it; rows past Out / batch repeat the last one and are not stored.  A macro: a lambda
    // that captures the four registers by reference puts them in scratch)
#define DENSE_FETCH(k0_)                                                                   \
    do {                                                                                   \
        const bool in_ = (k0_) + scol < In;                                                \
        const int kk_ = in_ ? (k0_) + scol : 0; /* unconditional loads: a select between a load and a constant took the constant's ADDRESS (scratch) */ \
        pw0 = *reinterpret_cast<const float4*>(wsrc0 + kk_);                               \
        pw1 = *reinterpret_cast<const float4*>(wsrc1 + kk_);                               \
        px0 = *reinterpret_cast<const float4*>(xsrc0 + kk_);                               \
        px1 = *reinterpret_cast<const float4*>(xsrc1 + kk_);                               \
        if (!in_) pw0 = pw1 = px0 = px1 = make_float4(0.f, 0.f, 0.f, 0.f);                 \
    } while (0)
    f32x16d acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    DENSE_FETCH(0);
    const float* const aop = tileW + r * kDensePitch + 16 * wave + 4 * kh; // this wave's K sixteenth, the lane's half of each 8
    const float* const bop = tileX + r * kDensePitch + 16 * wave + 4 * kh;
    for (int k0 = 0; k0 < In; k0 += kDenseKC) {
        if (k0) __syncthreads(); // every wave is done with the previous chunk
        *reinterpret_cast<float4*>(tileW + srow * kDensePitch + scol) = pw0;
        *reinterpret_cast<float4*>(tileW + (srow + 16) * kDensePitch + scol) = pw1;
        *reinterpret_cast<float4*>(tileX + srow * kDensePitch + scol) = px0;
        *reinterpret_cast<float4*>(tileX + (srow + 16) * kDensePitch + scol) = px1;
        __syncthreads();
        if (k0 + kDenseKC < In) DENSE_FETCH(k0 + kDenseKC);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 a = *reinterpret_cast<const float4*>(aop + 8 * j), b = *reinterpret_cast<const float4*>(bop + 8 * j);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    // D layout: lane (column = image b0 + l % 32, half h) holds rows (output units) 8 (i / 4) + 4 h + i % 4
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, i = e >> 6, ln = e & 63;
        const int b = b0 + (ln & 31), o = o0 + 8 * (i >> 2) + 4 * (ln >> 5) + (i & 3);
        if (b < batch && o < Out) {
            float v = ((red[0][i][ln] + red[1][i][ln]) + (red[2][i][ln] + red[3][i][ln])) + bias[o];
            switch (act) {
            case SNNHIP_DENSE_RELU: v = v > 0 ? v : 0.0f * v; break;
            case SNNHIP_DENSE_LEAKY: v = v > 0 ? v : leaky * v; break;
            case SNNHIP_DENSE_SIGMOID: v = 1.0f / (1.0f + expf(-v)); break;
            case SNNHIP_DENSE_TANH: v = (expf(2 * v) - 1) / (expf(2 * v) + 1); break;
            default: break;
            }
            y[static_cast<size_t>(b) * Out + o] = v;
        }
    }
#undef DENSE_FETCH
}

// softmax over one output row per block (cpulayer.h:173-189): max, exp(x-max), sum, divide
__global__ __launch_bounds__(256) void softmax_rows_kernel(int Out, float* __restrict__ y) {
    __shared__ float red[4];
    float* row = y + static_cast<size_t>(blockIdx.x) * Out;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float m = -3.402823466e+38f;
    for (int i = threadIdx.x; i < Out; i += 256) m = fmaxf(m, row[i]);
    m = wave_max(m);
    if (lane == 0) red[wv] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.0f;
    for (int i = threadIdx.x; i < Out; i += 256) {
        const float e = expf(row[i] - m);
        row[i] = e;
        s += e;
    }
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    s = red[0] + red[1] + red[2] + red[3];
    for (int i = threadIdx.x; i < Out; i += 256) row[i] = row[i] / s;
}

__global__ __launch_bounds__(256) void convert_rows_kernel(size_t n, const float* __restrict__ src, _Float16* __restrict__ dst) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256) dst[i] = static_cast<_Float16>(src[i]);
}

struct DensePlan : snnhip_plan {
    snnhip_dense_desc d;
    float* d_w = nullptr;
    float* d_b = nullptr;
    float* d_row = nullptr; // fp32 result rows when the output tensor holds halfs
    bool mfma = false;      // batch >= 32, In % 8 == 0 (fp32 tensors): dense_mfma_kernel

    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "dense: expects 1 input, got %d", nIn);
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == d.batch && static_cast<size_t>(x->h) * x->w * x->c == static_cast<size_t>(d.in_units),
                       "dense: input %dx%dx%dx%d does not flatten to %d x %d", x->n, x->h, x->w, x->c, d.batch, d.in_units);
        SNNHIP_REQUIRE(out->count() == static_cast<size_t>(d.batch) * d.out_units, "dense: output has %zu elements, expected %d x %d", out->count(),
                       d.batch, d.out_units);
        SNNHIP_REQUIRE(x->dtype == out->dtype, "dense: input dtype %d, output dtype %d", x->dtype, out->dtype);
        const int gb = d.batch >= 64 ? 16 : 4;
        dim3 grid(up_div(d.out_units, 4), up_div(d.batch, gb));
        const bool vec = (d.in_units % 4) == 0;
        const bool half = out->dtype == SNNHIP_F16;
        float* rows = half ? d_row : out->data;
        const bool useMfma = !half && mfma; // batched fp32 heads: the 32 x 32-tile MFMA GEMM
        if (half) {
            const _Float16* xh = reinterpret_cast<const _Float16*>(x->data);
#define SNNHIP_DENSE(V, TXX, XP)                                                                                                                          \
    do {                                                                                                                                             \
        if (gb == 16) SNNHIP_LAUNCH((dense_kernel<V, TXX, 16>), grid, dim3(256), 0, ctx->stream, d.in_units, d.out_units, d.batch, d.act, d.leaky, XP, d_w, d_b, rows); \
        else SNNHIP_LAUNCH((dense_kernel<V, TXX, 4>), grid, dim3(256), 0, ctx->stream, d.in_units, d.out_units, d.batch, d.act, d.leaky, XP, d_w, d_b, rows);           \
    } while (0)
            if (vec) SNNHIP_DENSE(true, _Float16, xh);
            else SNNHIP_DENSE(false, _Float16, xh);
        } else if (useMfma) {
            SNNHIP_LAUNCH(dense_mfma_kernel, dim3(up_div(d.out_units, 32), up_div(d.batch, 32)), dim3(256), 0, ctx->stream, d.in_units, d.out_units, d.batch, d.act, d.leaky,
                               x->data, d_w, d_b, rows);
        } else {
            if (vec) SNNHIP_DENSE(true, float, x->data);
            else SNNHIP_DENSE(false, float, x->data);
        }
#undef SNNHIP_DENSE
        SNNHIP_CHECK_HIP(hipGetLastError());
        if (d.act == SNNHIP_DENSE_SOFTMAX) {
            SNNHIP_LAUNCH(softmax_rows_kernel, dim3(d.batch), dim3(256), 0, ctx->stream, d.out_units, rows);
            SNNHIP_CHECK_HIP(hipGetLastError());
        }
        if (half) {
            const size_t n = static_cast<size_t>(d.batch) * d.out_units;
            SNNHIP_LAUNCH(convert_rows_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ctx->stream, n, rows,
                               reinterpret_cast<_Float16*>(out->data));
            SNNHIP_CHECK_HIP(hipGetLastError());
        }
        return SNNHIP_OK;
    }
};

// T = float or _Float16 storage (the reference's RGBA16F path: ESPCN with preferHp); tanh in fp32, round-to-nearest store
template <typename T>
__global__ __launch_bounds__(256) void subpixel_kernel(int N, int H, int W, int C, int f, int mode, const T* __restrict__ x, T* __restrict__ y) {
    const int OH = H * f, OW = W * f;
    const size_t total = static_cast<size_t>(N) * OH * OW;
    const int depth = (C + 3) / 4;
    for (size_t idx = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; idx < total; idx += static_cast<size_t>(gridDim.x) * 256) {
        const int ox = static_cast<int>(idx % OW);
        const size_t r = idx / OW;
        const int oy = static_cast<int>(r % OH);
        const int n = static_cast<int>(r / OH);
        const int x1 = min(ox / f, W - 1), y1 = min(oy / f, H - 1);
        const int z1 = (ox % f) + (oy % f) * f;
        const int ch = (mode == SNNHIP_SUBPIXEL_VK_QUIRK) ? min(z1, depth - 1) * 4 : z1;
        const float v = ch < C ? static_cast<float>(x[((static_cast<size_t>(n) * H + y1) * W + x1) * C + ch]) : 0.0f;
        y[idx] = static_cast<T>(tanhf(v));
    }
}

struct SubpixelPlan : SubpixelPlanBase {
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "subpixel: expects 1 input, got %d", nIn);
        SNNHIP_SAME_DTYPE("subpixel");
        const snnhip_tensor* x = in[0];
        SNNHIP_REQUIRE(x->n == d.N && x->h == d.H && x->w == d.W && x->c == d.C, "subpixel: input dims %dx%dx%dx%d != plan %dx%dx%dx%d", x->n, x->h,
                       x->w, x->c, d.N, d.H, d.W, d.C);
        SNNHIP_REQUIRE(out->n == d.N && out->h == d.H * d.factor && out->w == d.W * d.factor && out->c == 1,
                       "subpixel: output dims %dx%dx%dx%d != %dx%dx%dx1", out->n, out->h, out->w, out->c, d.N, d.H * d.factor, d.W * d.factor);
        const size_t total = out->count();
        size_t blocks = (total + 255) / 256;
        const size_t cap = static_cast<size_t>(ctx->props.multiProcessorCount) * 16;
        if (blocks > cap) blocks = cap;
        if (blocks == 0) return SNNHIP_OK;
        SNNHIP_WITH_T(out->dtype, SNNHIP_LAUNCH((subpixel_kernel<T>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, ctx->stream, d.N, d.H, d.W,
                                                     d.C, d.factor, d.mode, cptr<T>(x), mptr<T>(out)););
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

} // namespace

int make_dense_plan(snnhip_ctx* ctx, const snnhip_dense_desc& d, const float* w_flat, const float* bias, snnhip_plan** out) {
    auto* plan = new DensePlan();
    plan->ctx = ctx;
    plan->d = d;
    std::vector<float> b(static_cast<size_t>(d.out_units), 0.0f);
    if (d.useBias && bias) b.assign(bias, bias + d.out_units);
    int rc = plan->upload(w_flat, static_cast<size_t>(d.in_units) * d.out_units, &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(b.data(), b.size(), &plan->d_b);
    if (rc == SNNHIP_OK) {
        std::vector<float> z(static_cast<size_t>(d.batch) * d.out_units, 0.0f);
        rc = plan->upload(z.data(), z.size(), &plan->d_row);
    }
    plan->anyDtype = true; // activations may be fp32 or half (weights and arithmetic are fp32 either way)
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    plan->inDims[0] = d.batch; plan->inDims[1] = 1; plan->inDims[2] = 1; plan->inDims[3] = d.in_units;
    plan->outDims[0] = d.batch; plan->outDims[1] = 1; plan->outDims[2] = 1; plan->outDims[3] = d.out_units;
    plan->flops = 2.0 * d.batch * static_cast<double>(d.in_units) * d.out_units;
    plan->bytes = 4.0 * (static_cast<double>(d.in_units) * d.out_units + static_cast<double>(d.batch) * (d.in_units + d.out_units) + d.out_units);
    const char* dm = snnhip::option("SNNHIP_DENSE_MFMA");
    plan->mfma = d.batch >= 32 && d.in_units % 8 == 0 && !(dm && atoi(dm) == 0);
    char buf[200];
    snprintf(buf, sizeof(buf), "dense_f32 %s in=%d out=%d batch=%d act=%d", plan->mfma ? "mfma_f32_32x32x2 GEMM (32x32 tiles, K over 4 waves; half tensors: wave-per-row)" : "wave-per-row",
             d.in_units, d.out_units, d.batch, d.act);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

int make_subpixel_plan(snnhip_ctx* ctx, const snnhip_subpixel_desc& d, snnhip_plan** out) {
    auto* plan = new SubpixelPlan();
    plan->ctx = ctx;
    plan->d = d;
    plan->inDims[0] = d.N; plan->inDims[1] = d.H; plan->inDims[2] = d.W; plan->inDims[3] = d.C;
    plan->outDims[0] = d.N; plan->outDims[1] = d.H * d.factor; plan->outDims[2] = d.W * d.factor; plan->outDims[3] = 1;
    plan->flops = 0;
    plan->bytes = 4.0 * (static_cast<double>(d.N) * d.H * d.W * d.C + static_cast<double>(d.N) * d.H * d.W * d.factor * d.factor);
    char buf[128];
    snprintf(buf, sizeof(buf), "subpixel f=%d mode=%d c=%d", d.factor, d.mode, d.C);
    plan->desc = buf;
    plan->anyDtype = true; // element type taken from the tensors of the call (fp32 or half)
    *out = plan;
    return SNNHIP_OK;
}

} // namespace snnhip
