// shape_ops.hip -- SURVEY.md section 8(f) rank 4: the remaining graph operators (Concatenate, Unary, Conv2DTranspose, Calculate) and the
// device-side step either side of a model run (input resize + normalise, 8-bit image -> tensor, argmax of the classifier output).
// All but the transposed convolution are HBM-bound NHWC sweeps: 16-byte (fp32) / 8-byte (fp16) channel-contiguous accesses whenever the
// channel counts are multiples of 4, grid-stride loops; the element type comes from the tensors of each call.
//
// Replaces (reference): shadertemplate_vk_concat.comp:39-52 + concatenationVulkan.cpp:31-88, shadertemplate_vk_unary.comp:40-90 +
// unaryVulkan.cpp:30-83, shadertemplate_cs_4x_deconv_2s_RGBA.glsl:150-195 + deconv2dGL.cpp:282-355, shadertemplate_fs_calculation.glsl:25-41,
// shadertemplate_vk_resize.comp:41-62 + imageTextureVulkan.cpp:137-183, image.cpp:712-796 (norm2rgba32f), core.cpp:228-234 (argmax).
#include "epilogue.h"
#include "plan_util.h"
#include "snnhip_internal.h"

namespace snnhip {
namespace {

// ------------------------------------------------------------------------------------------------ concat
// One thread = one output texel (4 channels of one pixel).  P0 = ceil(C0/4): output plane p < P0 reads input 0, else input 1 (plane p - P0).
template <bool VEC, typename T>
__global__ __launch_bounds__(256) void concat_kernel(size_t pixels, int C0, int C1, int OC, const T* __restrict__ x0, const T* __restrict__ x1,
                                                    T* __restrict__ y) {
    const int P0 = (C0 + 3) / 4;
    const int OP = (OC + 3) / 4;
    const size_t total = pixels * OP;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const size_t px = i / OP;
        const int p = static_cast<int>(i - px * OP);
        const bool first = p < P0;
        const T* src = first ? x0 + px * C0 : x1 + px * C1;
        const int sc = first ? C0 : C1;
        const int c0 = 4 * (first ? p : p - P0);
        if (VEC) {
            float v[4];
            ldv<T, 4>(src + c0, v);
            stv<T, 4>(y + px * OC + 4 * p, v);
        } else {
            for (int l = 0; l < 4 && 4 * p + l < OC; ++l) y[px * OC + 4 * p + l] = (c0 + l < sc) ? src[c0 + l] : static_cast<T>(0.0f);
        }
    }
}

// ------------------------------------------------------------------------------------------------ unary
__device__ __forceinline__ float unary1(int op, float value, float v) {
    switch (op) {
    case SNNHIP_UNARY_FIXED: return value;
    case SNNHIP_UNARY_NEG: return 0.0f - v;
    case SNNHIP_UNARY_RCP: return 1.0f / v;
    case SNNHIP_UNARY_SQUARE: return v * v;
    case SNNHIP_UNARY_EXP: return expf(v);
    case SNNHIP_UNARY_ABS: return fabsf(v);
    default: return v;
    }
}

template <int CV, typename T>
__global__ __launch_bounds__(256) void unary_kernel(size_t count, int op, float value, const T* __restrict__ x, T* __restrict__ y) {
    const size_t ng = count / CV;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < ng; i += static_cast<size_t>(gridDim.x) * 256) {
        float v[CV];
        ldv<T, CV>(x + i * CV, v);
#pragma unroll
        for (int k = 0; k < CV; ++k) v[k] = unary1(op, value, v[k]);
        stv<T, CV>(y + i * CV, v);
    }
}

// ------------------------------------------------------------------------------------------------ calculate
template <typename T>
__global__ __launch_bounds__(256) void calculate_kernel(size_t pixels, int C, int OC, const T* __restrict__ x, T* __restrict__ y) {
    const size_t total = pixels * OC;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const size_t px = i / OC;
        const int l = static_cast<int>(i - px * OC) & 3;
        const float d = static_cast<float>(x[px * C + 8]);
        y[i] = static_cast<T>(l < 3 ? static_cast<float>(x[px * C + l]) / d : 0.0f);
    }
}

// ------------------------------------------------------------------------------------------------ resize + normalise
struct ResizeArgs {
    int N, H, W, C, OH, OW, linear;
    float means[4], norms[4];
};

template <int CV, typename T>
__global__ __launch_bounds__(256) void resize_kernel(ResizeArgs a, const T* __restrict__ x, T* __restrict__ y) {
    const int cg = a.C / CV;
    const size_t total = static_cast<size_t>(a.N) * a.OH * a.OW * cg;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const int c = static_cast<int>(i % cg) * CV;
        size_t r = i / cg;
        const int ox = static_cast<int>(r % a.OW);
        r /= a.OW;
        const int oy = static_cast<int>(r % a.OH);
        const int n = static_cast<int>(r / a.OH);
        // vk_resize.comp:50-57: normalised coordinate of the output texel centre, then the sampler's unnormalised position
        const float u = (static_cast<float>(ox) + 0.5f) / static_cast<float>(a.OW) * static_cast<float>(a.W);
        const float v = (static_cast<float>(oy) + 0.5f) / static_cast<float>(a.OH) * static_cast<float>(a.H);
        const T* img = x + static_cast<size_t>(n) * a.H * a.W * a.C;
        float o[CV];
        if (a.linear) {
            const float fu = u - 0.5f, fv = v - 0.5f;
            const float x0f = floorf(fu), y0f = floorf(fv);
            const float ax = fu - x0f, ay = fv - y0f;
            const int x0 = min(max(static_cast<int>(x0f), 0), a.W - 1), x1 = min(max(static_cast<int>(x0f) + 1, 0), a.W - 1);
            const int y0 = min(max(static_cast<int>(y0f), 0), a.H - 1), y1 = min(max(static_cast<int>(y0f) + 1, 0), a.H - 1);
            float p00[CV], p01[CV], p10[CV], p11[CV];
            ldv<T, CV>(img + (static_cast<size_t>(y0) * a.W + x0) * a.C + c, p00);
            ldv<T, CV>(img + (static_cast<size_t>(y0) * a.W + x1) * a.C + c, p01);
            ldv<T, CV>(img + (static_cast<size_t>(y1) * a.W + x0) * a.C + c, p10);
            ldv<T, CV>(img + (static_cast<size_t>(y1) * a.W + x1) * a.C + c, p11);
#pragma unroll
            for (int k = 0; k < CV; ++k) {
                const float top = p00[k] * (1.0f - ax) + p01[k] * ax;
                const float bot = p10[k] * (1.0f - ax) + p11[k] * ax;
                o[k] = top * (1.0f - ay) + bot * ay;
            }
        } else {
            const int xi = min(max(static_cast<int>(floorf(u)), 0), a.W - 1);
            const int yi = min(max(static_cast<int>(floorf(v)), 0), a.H - 1);
            ldv<T, CV>(img + (static_cast<size_t>(yi) * a.W + xi) * a.C + c, o);
        }
#pragma unroll
        for (int k = 0; k < CV; ++k) o[k] = (o[k] - a.means[(c + k) & 3]) * a.norms[(c + k) & 3];
        stv<T, CV>(y + i * CV, o);
    }
}

// ------------------------------------------------------------------------------------------------ 8-bit image -> tensor
template <typename T>
__global__ __launch_bounds__(256) void image_u8_kernel(size_t pixels, int sc, float4 means, float4 norms, const unsigned char* __restrict__ x, T* __restrict__ y) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < pixels; i += static_cast<size_t>(gridDim.x) * 256) {
        const unsigned char* s = x + i * sc;
        float v[4];
        if (sc == 4) {
            const uchar4 t = *reinterpret_cast<const uchar4*>(s);
            v[0] = (static_cast<float>(t.x) - means.x) * norms.x;
            v[1] = (static_cast<float>(t.y) - means.y) * norms.y;
            v[2] = (static_cast<float>(t.z) - means.z) * norms.z;
            v[3] = (static_cast<float>(t.w) - means.w) * norms.w;
        } else if (sc == 3) {
            v[0] = (static_cast<float>(s[0]) - means.x) * norms.x;
            v[1] = (static_cast<float>(s[1]) - means.y) * norms.y;
            v[2] = (static_cast<float>(s[2]) - means.z) * norms.z;
            v[3] = 1.0f; // image.cpp:744
        } else {
            v[0] = (static_cast<float>(s[0]) - means.x) * norms.x;
            v[1] = v[2] = v[3] = (0.0f - means.x) * norms.x; // image.cpp:748-750
        }
        stv<T, 4>(y + i * 4, v);
    }
}

// ------------------------------------------------------------------------------------------------ argmax
template <typename T>
__global__ __launch_bounds__(256) void argmax_kernel(size_t count, const T* __restrict__ x, int* __restrict__ out) {
    __shared__ float sv[256];
    __shared__ int si[256];
    float best = -__builtin_huge_valf();
    int idx = 0x7fffffff;
    for (size_t i = threadIdx.x; i < count; i += 256) {
        const float v = static_cast<float>(x[i]);
        if (v > best || idx == 0x7fffffff) { // strict: the first of equal elements wins, like std::max_element
            best = v;
            idx = static_cast<int>(i);
        }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = idx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (static_cast<int>(threadIdx.x) < s) {
            const float ov = sv[threadIdx.x + s];
            const int oi = si[threadIdx.x + s];
            if (oi != 0x7fffffff && (si[threadIdx.x] == 0x7fffffff || ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x]))) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = si[0];
}

// ------------------------------------------------------------------------------------------------ transposed convolution
// One thread = one output pixel x 4 output channels.  Consecutive threads take consecutive channel groups of the same pixel, so the
// activation loads are wave-uniform broadcasts and the weight loads ([tap][ic][OCg] float4) are contiguous across the wave.
struct DeconvArgs {
    int N, H, W, IC, OC, k, s, p, OH, OW, act, useBN;
    float leaky;
};

template <typename T>
__global__ __launch_bounds__(256) void deconv2d_kernel(DeconvArgs a, const T* __restrict__ x, const float4* __restrict__ w, const float4* __restrict__ epi,
                                                      T* __restrict__ y) {
    const int OCg = (a.OC + 3) / 4;
    const size_t total = static_cast<size_t>(a.N) * a.OH * a.OW * OCg;
    const int base = a.k - 1 - a.p;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const int og = static_cast<int>(i % OCg);
        size_t r = i / OCg;
        const int ox = static_cast<int>(r % a.OW);
        r /= a.OW;
        const int oy = static_cast<int>(r % a.OH);
        const int n = static_cast<int>(r / a.OH);
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        // taps with ky = base - oy + s*iy in [0, k)
        const int iy0 = max(0, (oy - base + a.s - 1 + a.s * a.k) / a.s - a.k), iy1 = min(a.H - 1, (oy - base + a.k - 1 + a.s * a.k) / a.s - a.k);
        const int ix0 = max(0, (ox - base + a.s - 1 + a.s * a.k) / a.s - a.k), ix1 = min(a.W - 1, (ox - base + a.k - 1 + a.s * a.k) / a.s - a.k);
        for (int iy = iy0; iy <= iy1; ++iy) {
            const int ky = base - oy + a.s * iy;
            for (int ix = ix0; ix <= ix1; ++ix) {
                const int kx = base - ox + a.s * ix;
                const T* px = x + ((static_cast<size_t>(n) * a.H + iy) * a.W + ix) * a.IC;
                const float4* wt = w + static_cast<size_t>(ky * a.k + kx) * a.IC * OCg + og;
                for (int ic = 0; ic < a.IC; ++ic) {
                    const float v = static_cast<float>(px[ic]);
                    const float4 q = wt[static_cast<size_t>(ic) * OCg];
                    acc.x = fmaf(v, q.x, acc.x);
                    acc.y = fmaf(v, q.y, acc.y);
                    acc.z = fmaf(v, q.z, acc.z);
                    acc.w = fmaf(v, q.w, acc.w);
                }
            }
        }
        const float av[4] = {acc.x, acc.y, acc.z, acc.w};
        T* dst = y + ((static_cast<size_t>(n) * a.OH + oy) * a.OW + ox) * a.OC + 4 * og;
        float o[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) o[l] = epi_act(a.act, a.leaky, epi_affine(av[l], epi[4 * og + l], a.useBN), 0.0f);
        if ((a.OC & 3) == 0) {
            stv<T, 4>(dst, o);
        } else {
            for (int l = 0; l < 4 && 4 * og + l < a.OC; ++l) dst[l] = static_cast<T>(o[l]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ plans
void set_dims(snnhip_plan* p, int n, int h, int w, int c, int oh, int ow, int oc) {
    p->inDims[0] = n; p->inDims[1] = h; p->inDims[2] = w; p->inDims[3] = c;
    p->outDims[0] = n; p->outDims[1] = oh; p->outDims[2] = ow; p->outDims[3] = oc;
}

struct ConcatPlan : snnhip_plan {
    snnhip_concat_desc d;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 2, "concat: expects 2 inputs, got %d", nIn);
        SNNHIP_SAME_DTYPE("concat");
        SNNHIP_REQUIRE(dims_match(in[0], d.N, d.H, d.W, d.C0) && dims_match(in[1], d.N, d.H, d.W, d.C1) && dims_match(out, d.N, d.H, d.W, d.OC),
                       "concat: tensor dims do not match the plan (%s)", desc.c_str());
        const size_t pixels = static_cast<size_t>(d.N) * d.H * d.W;
        const bool vec = ((d.C0 | d.C1 | d.OC) & 3) == 0 && d.OC <= d.C0 + d.C1;
        const unsigned g = grid_for(ctx, pixels * ((d.OC + 3) / 4));
        SNNHIP_WITH_T(out->dtype,
                      if (vec) SNNHIP_LAUNCH((concat_kernel<true, T>), dim3(g), dim3(256), 0, ctx->stream, pixels, d.C0, d.C1, d.OC, cptr<T>(in[0]), cptr<T>(in[1]), mptr<T>(out));
                      else SNNHIP_LAUNCH((concat_kernel<false, T>), dim3(g), dim3(256), 0, ctx->stream, pixels, d.C0, d.C1, d.OC, cptr<T>(in[0]), cptr<T>(in[1]), mptr<T>(out)););
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

struct UnaryPlan : snnhip_plan {
    snnhip_unary_desc d;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "unary: expects 1 input, got %d", nIn);
        SNNHIP_SAME_DTYPE("unary");
        SNNHIP_REQUIRE(dims_match(in[0], d.N, d.H, d.W, d.C) && dims_match(out, d.N, d.H, d.W, d.C), "unary: tensor dims do not match the plan");
        const size_t count = out->count();
        const bool vec = (d.C & 3) == 0;
        const unsigned g = grid_for(ctx, vec ? count / 4 : count);
        SNNHIP_WITH_T(out->dtype, if (vec) SNNHIP_LAUNCH((unary_kernel<4, T>), dim3(g), dim3(256), 0, ctx->stream, count, d.op, d.value, cptr<T>(in[0]), mptr<T>(out));
                      else SNNHIP_LAUNCH((unary_kernel<1, T>), dim3(g), dim3(256), 0, ctx->stream, count, d.op, d.value, cptr<T>(in[0]), mptr<T>(out)););
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

struct CalculatePlan : snnhip_plan {
    snnhip_calculate_desc d;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "calculate: expects 1 input, got %d", nIn);
        SNNHIP_SAME_DTYPE("calculate");
        SNNHIP_REQUIRE(dims_match(in[0], d.N, d.H, d.W, d.C) && dims_match(out, d.N, d.H, d.W, d.OC), "calculate: tensor dims do not match the plan");
        const size_t pixels = static_cast<size_t>(d.N) * d.H * d.W;
        const unsigned g = grid_for(ctx, pixels * d.OC);
        SNNHIP_WITH_T(out->dtype, SNNHIP_LAUNCH((calculate_kernel<T>), dim3(g), dim3(256), 0, ctx->stream, pixels, d.C, d.OC, cptr<T>(in[0]), mptr<T>(out)););
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

struct ResizePlan : snnhip_plan {
    ResizeArgs a;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "resize: expects 1 input, got %d", nIn);
        SNNHIP_SAME_DTYPE("resize");
        SNNHIP_REQUIRE(dims_match(in[0], a.N, a.H, a.W, a.C) && dims_match(out, a.N, a.OH, a.OW, a.C), "resize: tensor dims do not match the plan");
        const bool vec = (a.C & 3) == 0;
        const unsigned g = grid_for(ctx, out->count() / (vec ? 4 : 1));
        SNNHIP_WITH_T(out->dtype, if (vec) SNNHIP_LAUNCH((resize_kernel<4, T>), dim3(g), dim3(256), 0, ctx->stream, a, cptr<T>(in[0]), mptr<T>(out));
                      else SNNHIP_LAUNCH((resize_kernel<1, T>), dim3(g), dim3(256), 0, ctx->stream, a, cptr<T>(in[0]), mptr<T>(out)););
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

struct ImageU8Plan : snnhip_plan {
    snnhip_image_u8_desc d;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "image_u8: expects 1 input, got %d", nIn);
        SNNHIP_REQUIRE(in[0]->dtype == SNNHIP_U8, "image_u8: the input tensor must be SNNHIP_U8, got dtype %d", in[0]->dtype);
        SNNHIP_REQUIRE(out->dtype == SNNHIP_F32 || out->dtype == SNNHIP_F16, "image_u8: output dtype %d", out->dtype);
        SNNHIP_REQUIRE(dims_match(in[0], d.N, d.H, d.W, d.src_channels) && dims_match(out, d.N, d.H, d.W, 4), "image_u8: tensor dims do not match the plan");
        const size_t pixels = static_cast<size_t>(d.N) * d.H * d.W;
        const float4 m = make_float4(d.means[0], d.means[1], d.means[2], d.means[3]), s = make_float4(d.norms[0], d.norms[1], d.norms[2], d.norms[3]);
        const unsigned g = grid_for(ctx, pixels);
        const unsigned char* src = reinterpret_cast<const unsigned char*>(in[0]->data);
        SNNHIP_WITH_T(out->dtype, SNNHIP_LAUNCH((image_u8_kernel<T>), dim3(g), dim3(256), 0, ctx->stream, pixels, d.src_channels, m, s, src, mptr<T>(out)););
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

struct DeconvPlan : snnhip_plan {
    DeconvArgs a;
    float* d_w = nullptr;
    float* d_epi = nullptr;
    int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) override {
        SNNHIP_REQUIRE(nIn == 1, "deconv2d: expects 1 input, got %d", nIn);
        SNNHIP_SAME_DTYPE("deconv2d");
        SNNHIP_REQUIRE(dims_match(in[0], a.N, a.H, a.W, a.IC) && dims_match(out, a.N, a.OH, a.OW, a.OC), "deconv2d: tensor dims do not match the plan (%s)",
                       desc.c_str());
        const unsigned g = grid_for(ctx, static_cast<size_t>(a.N) * a.OH * a.OW * ((a.OC + 3) / 4));
        SNNHIP_WITH_T(out->dtype, SNNHIP_LAUNCH((deconv2d_kernel<T>), dim3(g), dim3(256), 0, ctx->stream, a, cptr<T>(in[0]), reinterpret_cast<const float4*>(d_w),
                                                     reinterpret_cast<const float4*>(d_epi), mptr<T>(out)););
        SNNHIP_CHECK_HIP(hipGetLastError());
        return SNNHIP_OK;
    }
};

} // namespace
} // namespace snnhip

using namespace snnhip;

extern "C" {

int snnhip_concat_plan_create(snnhip_ctx* ctx, const snnhip_concat_desc* desc, snnhip_plan** out) {
    SNNHIP_REQUIRE(ctx && desc && out, "concat_plan_create: null argument");
    SNNHIP_REQUIRE(desc->N > 0 && desc->H > 0 && desc->W > 0 && desc->C0 > 0 && desc->C1 > 0 && desc->OC > 0, "concat desc: bad dims");
    auto* plan = new ConcatPlan();
    plan->ctx = ctx;
    plan->anyDtype = true;
    plan->numInputs = 2;
    plan->d = *desc;
    set_dims(plan, desc->N, desc->H, desc->W, desc->C0, desc->H, desc->W, desc->OC);
    const double px = static_cast<double>(desc->N) * desc->H * desc->W;
    plan->bytes = 4.0 * px * (std::min(desc->OC, desc->C0 + desc->C1) + desc->OC);
    char buf[160];
    snprintf(buf, sizeof(buf), "concat c=%d+%d->%d %dx%d%s", desc->C0, desc->C1, desc->OC, desc->H, desc->W, ((desc->C0 | desc->C1 | desc->OC) & 3) ? " scalar" : " vec4");
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

int snnhip_unary_plan_create(snnhip_ctx* ctx, const snnhip_unary_desc* desc, snnhip_plan** out) {
    SNNHIP_REQUIRE(ctx && desc && out, "unary_plan_create: null argument");
    SNNHIP_REQUIRE(desc->N > 0 && desc->H > 0 && desc->W > 0 && desc->C > 0, "unary desc: bad dims");
    SNNHIP_REQUIRE(desc->op >= SNNHIP_UNARY_COPY && desc->op <= SNNHIP_UNARY_ABS, "unary desc: op %d", desc->op);
    auto* plan = new UnaryPlan();
    plan->ctx = ctx;
    plan->anyDtype = true;
    plan->d = *desc;
    set_dims(plan, desc->N, desc->H, desc->W, desc->C, desc->H, desc->W, desc->C);
    plan->bytes = 8.0 * desc->N * desc->H * desc->W * desc->C;
    char buf[128];
    snprintf(buf, sizeof(buf), "unary op=%d c=%d %dx%d", desc->op, desc->C, desc->H, desc->W);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

int snnhip_calculate_plan_create(snnhip_ctx* ctx, const snnhip_calculate_desc* desc, snnhip_plan** out) {
    SNNHIP_REQUIRE(ctx && desc && out, "calculate_plan_create: null argument");
    SNNHIP_REQUIRE(desc->N > 0 && desc->H > 0 && desc->W > 0 && desc->OC > 0, "calculate desc: bad dims");
    SNNHIP_REQUIRE(desc->C >= 9, "calculate desc: %d input channels, the divisor is channel 8 (fs_calculation.glsl:36)", desc->C);
    auto* plan = new CalculatePlan();
    plan->ctx = ctx;
    plan->anyDtype = true;
    plan->d = *desc;
    set_dims(plan, desc->N, desc->H, desc->W, desc->C, desc->H, desc->W, desc->OC);
    plan->bytes = 4.0 * desc->N * desc->H * desc->W * (4 + desc->OC);
    char buf[128];
    snprintf(buf, sizeof(buf), "calculate c=%d->%d %dx%d", desc->C, desc->OC, desc->H, desc->W);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

int snnhip_resize_plan_create(snnhip_ctx* ctx, const snnhip_resize_desc* desc, snnhip_plan** out) {
    SNNHIP_REQUIRE(ctx && desc && out, "resize_plan_create: null argument");
    SNNHIP_REQUIRE(desc->N > 0 && desc->H > 0 && desc->W > 0 && desc->C > 0 && desc->OH > 0 && desc->OW > 0, "resize desc: bad dims");
    auto* plan = new ResizePlan();
    plan->ctx = ctx;
    plan->anyDtype = true;
    ResizeArgs& a = plan->a;
    a.N = desc->N; a.H = desc->H; a.W = desc->W; a.C = desc->C; a.OH = desc->OH; a.OW = desc->OW; a.linear = desc->linear ? 1 : 0;
    for (int i = 0; i < 4; ++i) {
        a.means[i] = desc->means[i];
        a.norms[i] = desc->norms[i];
    }
    set_dims(plan, desc->N, desc->H, desc->W, desc->C, desc->OH, desc->OW, desc->C);
    plan->bytes = 4.0 * desc->N * desc->C * (static_cast<double>(desc->H) * desc->W + static_cast<double>(desc->OH) * desc->OW);
    char buf[128];
    snprintf(buf, sizeof(buf), "resize_%s c=%d %dx%d->%dx%d", a.linear ? "linear" : "nearest", desc->C, desc->H, desc->W, desc->OH, desc->OW);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

int snnhip_image_u8_plan_create(snnhip_ctx* ctx, const snnhip_image_u8_desc* desc, snnhip_plan** out) {
    SNNHIP_REQUIRE(ctx && desc && out, "image_u8_plan_create: null argument");
    SNNHIP_REQUIRE(desc->N > 0 && desc->H > 0 && desc->W > 0, "image_u8 desc: bad dims");
    SNNHIP_REQUIRE(desc->src_channels == 1 || desc->src_channels == 3 || desc->src_channels == 4, "image_u8 desc: %d source channels (R8, RGB8, RGBA8)",
                   desc->src_channels);
    auto* plan = new ImageU8Plan();
    plan->ctx = ctx;
    plan->anyDtype = true;
    plan->u8Input = true;
    plan->d = *desc;
    set_dims(plan, desc->N, desc->H, desc->W, desc->src_channels, desc->H, desc->W, 4);
    plan->bytes = static_cast<double>(desc->N) * desc->H * desc->W * (desc->src_channels + 16);
    char buf[128];
    snprintf(buf, sizeof(buf), "image_u8 ch=%d %dx%d", desc->src_channels, desc->H, desc->W);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

int snnhip_deconv2d_plan_create(snnhip_ctx* ctx, const snnhip_conv2d_desc* desc, const float* w_oihw, const float* bias, const float* bn_beta,
                                const float* bn_gamma, const float* bn_mean, const float* bn_var, snnhip_plan** out) {
    SNNHIP_REQUIRE(ctx && desc && w_oihw && out, "deconv2d_plan_create: null argument");
    SNNHIP_REQUIRE(desc->N > 0 && desc->H > 0 && desc->W > 0 && desc->IC > 0 && desc->OC > 0, "deconv2d desc: bad dims");
    SNNHIP_REQUIRE(desc->kh == desc->kw && desc->kh > 0 && desc->sh == desc->sw && desc->sh > 0, "deconv2d desc: square kernels and strides only (Conv2DDesc)");
    SNNHIP_REQUIRE(desc->padT >= 0 && desc->padT < desc->kh, "deconv2d desc: padT %d", desc->padT);
    SNNHIP_REQUIRE(desc->act == SNNHIP_ACT_NONE || desc->act == SNNHIP_ACT_RELU || desc->act == SNNHIP_ACT_TANH || desc->act == SNNHIP_ACT_SIGMOID ||
                       desc->act == SNNHIP_ACT_LEAKY,
                   "deconv2d desc: activation id %d (deconv2dGL.cpp:198-207 knows relu, tanh, sigmoid, leakyRelu)", desc->act);
    if (desc->useBN) SNNHIP_REQUIRE(bn_beta && bn_gamma && bn_mean && bn_var, "deconv2d_plan_create: useBN set but a BN array is null");
    SNNHIP_CHECK_HIP(hipSetDevice(ctx->device));
    auto* plan = new DeconvPlan();
    plan->ctx = ctx;
    plan->anyDtype = true;
    DeconvArgs& a = plan->a;
    a.N = desc->N; a.H = desc->H; a.W = desc->W; a.IC = desc->IC; a.OC = desc->OC; a.k = desc->kh; a.s = desc->sh; a.p = desc->padT;
    a.act = desc->act; a.leaky = desc->leaky; a.useBN = desc->useBN ? 1 : 0;
    a.OH = desc->OH > 0 ? desc->OH : desc->sh * desc->H; // deconv2dGL.cpp:345-355 ("same"); "valid" callers pass s*H + k - s
    a.OW = desc->OW > 0 ? desc->OW : desc->sw * desc->W;
    const int OCg = (a.OC + 3) / 4;
    std::vector<float> wp(static_cast<size_t>(a.k) * a.k * a.IC * OCg * 4, 0.0f);
    for (int o = 0; o < a.OC; ++o)
        for (int i = 0; i < a.IC; ++i)
            for (int t = 0; t < a.k * a.k; ++t)
                wp[((static_cast<size_t>(t) * a.IC + i) * OCg + o / 4) * 4 + (o & 3)] = w_oihw[(static_cast<size_t>(o) * a.IC + i) * a.k * a.k + t];
    std::vector<float> epi = make_epilogue_table(a.OC, 4, desc->useBias && bias, bias, desc->useBN, bn_beta, bn_gamma, bn_mean, bn_var);
    int rc = plan->upload(wp.data(), wp.size(), &plan->d_w);
    if (rc == SNNHIP_OK) rc = plan->upload(epi.data(), epi.size(), &plan->d_epi);
    if (rc != SNNHIP_OK) {
        delete plan;
        return rc;
    }
    set_dims(plan, a.N, a.H, a.W, a.IC, a.OH, a.OW, a.OC);
    // every input pixel meets (k/s)^2 taps per output pixel it feeds: k*k*IC*OC MACs per INPUT pixel (borders clip a few)
    plan->flops = 2.0 * a.N * a.H * a.W * a.k * a.k * static_cast<double>(a.IC) * a.OC;
    plan->bytes = 4.0 * (static_cast<double>(a.N) * a.H * a.W * a.IC + static_cast<double>(a.N) * a.OH * a.OW * a.OC + static_cast<double>(wp.size()));
    char buf[160];
    snprintf(buf, sizeof(buf), "deconv2d_direct k=%d s=%d p=%d ic=%d oc=%d %dx%d->%dx%d", a.k, a.s, a.p, a.IC, a.OC, a.H, a.W, a.OH, a.OW);
    plan->desc = buf;
    *out = plan;
    return SNNHIP_OK;
}

int snnhip_tensor_argmax(const snnhip_tensor* t, int n, int* out_index) {
    SNNHIP_REQUIRE(t && out_index && t->data, "tensor_argmax: null argument");
    SNNHIP_REQUIRE(n >= 0 && n < t->n, "tensor_argmax: image %d of %d", n, t->n);
    SNNHIP_REQUIRE(t->dtype == SNNHIP_F32 || t->dtype == SNNHIP_F16, "tensor_argmax: dtype %d", t->dtype);
    int* d = nullptr;
    SNNHIP_CHECK_HIP(snnhip::dev_malloc(&d, sizeof(int)));
    const size_t per = t->count() / t->n;
    if (t->dtype == SNNHIP_F16)
        SNNHIP_LAUNCH((argmax_kernel<_Float16>), dim3(1), dim3(256), 0, t->ctx->stream, per, reinterpret_cast<const _Float16*>(t->data) + per * n, d);
    else
        SNNHIP_LAUNCH((argmax_kernel<float>), dim3(1), dim3(256), 0, t->ctx->stream, per, t->data + per * n, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out_index, d, sizeof(int), hipMemcpyDeviceToHost, t->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->ctx->stream);
    (void) snnhip::dev_free(d);
    SNNHIP_CHECK_HIP(e);
    return SNNHIP_OK;
}

} // extern "C"
