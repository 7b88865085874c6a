/*
 * snn_oracle.h -- CPU restatement of ShaderNN's conv / depthwise / dense / subpixel operator arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the reported CPU baseline.  The HIP path (shadernn_amd/csrc) never links or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   dense      -- PINNED: checked against the reference's own Eigen path (core/src/ic2/cpulayer.h:136-171)
 *                 compiled from /root/reference into oracle/_ref (tests/golden/dense_ref_*.json).
 *   conv2d / depthwise / subpixel -- PARITY UNPINNED by reference fixtures: the reference has no CPU conv,
 *                 its test ground truth (ncnn 20211208, demo/install-deps.sh:18-25) is not vendored, and it
 *                 ships no golden vectors.  These functions restate the GLSL compute shaders line by line
 *                 (two independent walks: C4HW4 "texel" walk in shader loop order, and a plain NHWC walk) and
 *                 are cross-checked against torch CPU conv2d (the same math ncnn's naive layer computes).
 *
 * All tensors are fp32.  "NHWC" = [N][H][W][C] true-channel.  "C4HW4" = the reference's RGBA 3-D texture
 * layout [ceil(C/4)][H][W][4] (core/src/ic2/dp.cpp:328-332, demo/common/shaderUnitTest.cpp:87-129).
 */
#ifndef SNN_ORACLE_H
#define SNN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* activation ids = shader specialisation constant 15 (core/src/ic2/conv2dVulkan.cpp:58-72) */
enum {
    SNN_ACT_NONE = 0,
    SNN_ACT_RELU = 1,
    SNN_ACT_RELU6 = 2,
    SNN_ACT_TANH = 3,
    SNN_ACT_SIGMOID = 4,
    SNN_ACT_LEAKY = 5,
    SNN_ACT_SILU = 6,       /* mathematically correct SiLU: x*sigmoid(x) */
    SNN_ACT_SILU_QUIRK = 7  /* reference shader bug (shadertemplate_vk_conv2d.comp:336-339): pixels 1..3 of every
                               aligned 4-pixel x group are gated by sigmoid(silu(pixel 0)) instead of their own */
};

/* padding mode = specialisation constant 16 (conv2dVulkan.cpp:74-81) */
enum { SNN_PAD_NONE = 0, SNN_PAD_CONSTANT = 1, SNN_PAD_REPLICATE = 2, SNN_PAD_REFLECT = 3 };

typedef struct {
    int N, H, W, IC, OC;  /* input batch/height/width/channels, output channels            */
    int kh, kw, sh, sw;   /* kernel and stride (reference always passes kh==kw, sh==sw)    */
    int padT, padB, padL, padR; /* Conv2DLayer::getPaddingOffset order (conv2d.cpp:39-74)  */
    int padMode;          /* SNN_PAD_*                                                      */
    int act;              /* SNN_ACT_*                                                      */
    float leaky;          /* leakyReluVal                                                   */
    int useBias;          /* bias buffer is always read by the shaders; 0 => treat as zeros */
    int useBN;            /* fused batch-norm, eps 1e-3 hard coded (vk_conv2d.comp:277-288) */
    int OH, OW;           /* output size (fill with snn_oracle_out_dim)                     */
} snn_oracle_conv_desc;

/* ---- host-side rules ------------------------------------------------------------------- */

/* Conv2DLayer::getPaddingOffset (conv2d.cpp:39-74 / separableconvolution.cpp:27-62).
 * `padding` is the JSON string: "same" | "valid" | "none" | decimal digits. offsets = {T,B,L,R}. */
void snn_oracle_padding_offsets(const char* padding, int kernel, int offsets[4]);

/* Conv2DLayer::getOutputScaleDimAdjustment + GenericModelLayer::getOutputDims
 * (conv2d.cpp:102-113, genericlayer.cpp:64-90): out = (uint32)(in*(1/s) + [1 + (T+B[-1 if k even] - k)/s]),
 * evaluated in float exactly as the reference does. */
int snn_oracle_out_dim(int in, int kernel, int stride, int padT, int padB);

/* Conv2DLayer::oihw2hwo4i4 (conv2d.cpp:76-100). out has roundup(OC,4)*kw*kh*roundup(IC,4) floats. */
void snn_oracle_pack_conv_weights(const float* oihw, int IC, int OC, int kw, int kh, float* out);

/* SeparableConv2DLayer::oihw2hwo4i4 (separableconvolution.cpp:88-111). w_chw = [C][kh][kw];
 * out has roundup(C,4)*kw*kh floats. */
void snn_oracle_pack_depthwise_weights(const float* w_chw, int C, int kw, int kh, float* out);

/* hwcToC4 (demo/common/shaderUnitTest.cpp:87-129) and its inverse (drops pad channels). */
void snn_oracle_hwc_to_c4hw4(const float* hwc, int H, int W, int C, float* c4);
void snn_oracle_c4hw4_to_hwc(const float* c4, int H, int W, int C, float* hwc);

/* snn::convertToMediumPrecision (core/src/utils.cpp:127-174): fp32 -> fp16 by TRUNCATION -> fp32. */
float snn_oracle_to_medium_precision(float v);

/* ---- operators, plain NHWC walk ------------------------------------------------------------ */

/* w_oihw = [OC][IC][kh][kw]; bias = [OC] or NULL; bn = 4 arrays of OC: beta, gamma, mean, variance. */
void snn_oracle_conv2d_nhwc(const snn_oracle_conv_desc* d, const float* x, const float* w_oihw, const float* bias,
                            const float* bn_beta, const float* bn_gamma, const float* bn_mean, const float* bn_var,
                            float* y);

/* Same, split over `threads` pthreads by output rows (used only as bench.py's cpu_baseline). */
void snn_oracle_conv2d_nhwc_mt(const snn_oracle_conv_desc* d, const float* x, const float* w_oihw, const float* bias,
                               const float* bn_beta, const float* bn_gamma, const float* bn_mean, const float* bn_var,
                               float* y, int threads);

/* depthwise: IC==OC==C; w_chw = [C][kh][kw]. Padding is always zero-by-tap-clipping
 * (shadertemplate_vk_depthwise.comp:77-78; padMode is ignored like the shader does). */
void snn_oracle_depthwise_nhwc(const snn_oracle_conv_desc* d, const float* x, const float* w_chw, const float* bias,
                               const float* bn_beta, const float* bn_gamma, const float* bn_mean, const float* bn_var,
                               float* y);

/* ---- operators, C4HW4 texel walk in the GLSL loop order (second, independent statement) ------ */

/* shadertemplate_vk_conv2d.comp:148-347 (k>1) and shadertemplate_vk_conv2d_1x1.comp:68-210 (k==1),
 * chosen exactly as Conv2DLayerVulkan::createCS does (conv2dVulkan.cpp:154). batch 1 only.
 * w_packed from snn_oracle_pack_conv_weights; bias4/bn*4 are padded to roundup(OC,4) (zeros). */
void snn_oracle_conv2d_texel(const snn_oracle_conv_desc* d, const float* x_c4, const float* w_packed, const float* bias4,
                             const float* bn_beta4, const float* bn_gamma4, const float* bn_mean4, const float* bn_var4,
                             float* y_c4);

/* shadertemplate_vk_depthwise.comp:64-137 */
void snn_oracle_depthwise_texel(const snn_oracle_conv_desc* d, const float* x_c4, const float* w_packed, const float* bias4,
                                const float* bn_beta4, const float* bn_gamma4, const float* bn_mean4, const float* bn_var4,
                                float* y_c4);

/* ---- dense ------------------------------------------------------------------------------- */

/* CPU activation names of CPUCommonUtil (cpulayer.h:38-42,199-261). */
enum {
    SNN_DENSE_ACT_IDENTITY = 0,
    SNN_DENSE_ACT_RELU = 1,
    SNN_DENSE_ACT_LEAKY = 2,
    SNN_DENSE_ACT_SIGMOID = 3,
    SNN_DENSE_ACT_SOFTMAX = 4,
    SNN_DENSE_ACT_TANH = 5,   /* (e^{2x}-1)/(e^{2x}+1), cpulayer.h:195 */
    SNN_DENSE_ACT_SILU = 6    /* reference iterates by value => no-op (cpulayer.h:245-252) */
};

/* Maps an activation string the way CPUCommonUtil::activation does: unknown strings (e.g. "linear",
 * "relu6") hit std::unordered_map::operator[] and default-construct to RELU (cpulayer.h:38-42,200). */
int snn_oracle_dense_act_from_string(const char* s);

/* DenseLayer::computeImageTexture -> CPUCommonUtil::transform (cpulayer.h:136-171):
 * y[b][o] = act( sum_i w_flat[o*In + i] * x[b][i] + bias[o] ),  w_flat = the parser's flat "kernel" array. */
void snn_oracle_dense(const float* x, int batch, int In, int Out, const float* w_flat, const float* bias, int act, float leaky,
                      float* y);

/* CPU flatten (cpulayer.h:94-113): C4HW4 texture -> row,column,plane,channel order == HWC, pad channels dropped. */
void snn_oracle_flatten_c4hw4(const float* x_c4, int H, int W, int C, float* out_hwc);

/* ---- subpixel (ESPCN tail) ------------------------------------------------------------------ */

enum {
    SNN_SUBPIXEL_D2S = 0,        /* true depth-to-space(2) + tanh (Keras reference, fs_subpixel.glsl:41-64) */
    SNN_SUBPIXEL_VK_QUIRK = 1    /* shadertemplate_vk_subpixel.comp:57-66: z clamps to depth-slice, takes .x */
};
/* x = [N][H][W][C] (C == factor*factor for ESPCN), y = [N][H*f][W*f][1]. */
void snn_oracle_subpixel_nhwc(const float* x, int N, int H, int W, int C, int factor, int mode, float* y);

/* ---- deterministic generators shared by tests / bench (reference demo/common/prng.h, testutil.cpp:41-46) */
/* ---- element-wise / pooling / shape operators, NHWC ------------------------------- */
/* restatements of the Vulkan compute shaders, one output element at a time in the shader's own order of operations.
 * PARITY UNPINNED by reference fixtures (the reference tests them against ncnn layers on a GPU); cross-checked
 * against torch CPU ops in tests/test_oracle.py. */

/* vk_add.comp:41-88 / vk_activation.comp:41-86: y = act(a + b) (b may be NULL: y = act(a)) */
void snn_oracle_add_act(const float* a, const float* b, long count, int act, float leaky, float* y);
/* Add with inputs of different extent: output H x W = max over the inputs (genericlayer.cpp:64-90); the shader runs over the first
 * input's extent only and fetches of the second input outside its texture return 0 (addlayerVulkan.cpp:44-46, vk_add.comp:47-49);
 * the rest of the output texture is never written by the reference -- zeros here */
void snn_oracle_add_ragged(const float* a, int H0, int W0, const float* b, int H1, int W1, int N, int C, int act, float leaky, float* y);
/* vk_batchnorm.comp:54-104: y = act(gamma / max(sqrt(var + 1e-3), 1e-4) * (x - mean) + beta), channel = index % C */
void snn_oracle_batchnorm(const float* x, long pixels, int C, const float* beta, const float* gamma, const float* mean, const float* var, int act,
                          float leaky, float* y);
/* MaxPooling2DLayer / AveragePooling2DLayer::getOutputScaleDimAdjustment (maxpool2d.cpp:26-36, avgpool2d.cpp:20-29) through
 * GenericModelLayer::getOutputDims (genericlayer.cpp:64-90): trunc(in / stride + max(0, same ? 1 - 1/stride : 1 - k/stride)) */
int snn_oracle_pool_out_dim(int in, int kernel, int stride, int same);
/* vk_maxpool2d.comp:42-74 (type 0, starts at -100000.0) / vk_avgpool2d.comp:42-69 (type 1, divides by the clipped tap count);
 * padT / padL are what the host puts into uConstant.pad (the Vulkan layers force 0: maxpool2dVulkan.cpp:62-64) */
void snn_oracle_pool2d(const float* x, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int padT, int padL, int OH, int OW, int type,
                       float* y);
/* vk_pad.comp:42-71 with uPad = {padT, padL} as padlayerVulkan.cpp:81-82 passes it (x is shifted by padT, y by padL);
 * mode 0 constant (zeros), 1 replicate, 2 reflect; output (H+padT+padB) x (W+padL+padR) (padlayer.cpp:58-67) */
void snn_oracle_pad(const float* x, int N, int H, int W, int C, int padT, int padB, int padL, int padR, int mode, float* y);
/* vk_upsampling2d_nearest.comp:43-66 (mode 0) / vk_upsampling2d_bilinear.comp:43-76 (mode 1); the shader samples at
 * pos * (1/scale) (upsampling2dVulkan.cpp:101); output trunc(in * scale) (upsampling2d.h:41-44) */
void snn_oracle_upsample(const float* x, int N, int H, int W, int C, float scale, int mode, float* y);
/* vk_instancenorm.comp:53-160: per image and channel mean over H*W, biased variance around that mean, eps as given
 * (the shader hard-codes 1e-5), y = act((x - mean) * gamma / sqrt(var + eps) + beta) */
void snn_oracle_instancenorm(const float* x, int N, int H, int W, int C, const float* beta, const float* gamma, float eps, int act, float leaky,
                             float* y);

/* ---- SURVEY.md section 8(f) rank 4 ---- */
/* vk_concat.comp:39-52 with inImgDepths = {ceil(C0/4), ceil(C1/4)} (concatenationVulkan.cpp:50-57, dp.cpp:591-595): output TEXEL PLANE p
 * comes from input 0 while p < ceil(C0/4), else from input 1's plane p - ceil(C0/4); lanes past an input's channel count read the
 * texture's zero padding; the output keeps OC channels (ceil(OC/4) planes exist, writes past them are dropped) */
void snn_oracle_concat(const float* x0, const float* x1, long pixels, int C0, int C1, int OC, float* y);
/* vk_unary.comp:40-90: op 0 copy, 1 fixed value, 2 negate, 3 reciprocal, 4 square, 5 exp, 6 abs */
void snn_oracle_unary(const float* x, long count, int op, float value, float* y);
/* fs_calculation.glsl:25-41: y[c] = x[c % 4] / x[8] for c % 4 < 3, else 0 (every output pass of calculationGL.cpp:38-55 runs the same shader) */
void snn_oracle_calculate(const float* x, long pixels, int C, int OC, float* y);
/* vk_resize.comp:41-62 under the samplers of vulkanImageResizeOp.cpp:42-70 (linear, clamp to edge) / vulkanImageTransformShaderOp (nearest):
 * sample at ((x + 0.5) / OW * W, (y + 0.5) / OH * H), then (value - means[c % 4]) * norms[c % 4].  Linear = the Vulkan bilinear
 * definition with exact weights (hardware quantises them to 8 bits) */
void snn_oracle_resize(const float* x, int N, int H, int W, int C, int OH, int OW, const float* means4, const float* norms4, int linear, float* y);
/* image.cpp:712-751 (norm2rgba32f) for RGBA8 (sc 4), RGB8 (3, alpha = 1), R8 (1, the other lanes = (0 - means[0]) * norms[0]) */
void snn_oracle_image_u8(const unsigned char* x, long pixels, int sc, const float* means4, const float* norms4, float* y);
/* core.cpp:228-234: std::distance(begin, std::max_element(begin, end)) -- the first of equal maxima (the reference reports it + 1) */
long snn_oracle_argmax(const float* x, long count);
/* Transposed convolution, general form of the reference's k=4 s=2 compute shader (see snn_oracle_deconv4x4s2_shader):
 * y[oy][ox][o] = act(BN(bias[o] + sum x[iy][ix][i] * w[o][i][k-1-p - oy + s*iy][k-1-p - ox + s*ix])), zero outside the input, BN as
 * cs_4x_deconv_2s_RGBA.glsl:185-191 (gamma / sqrt(var + 1e-3)); bn arrays may be NULL */
void snn_oracle_deconv2d(const float* x, int N, int H, int W, int IC, int OC, int k, int s, int p, int OH, int OW, const float* w_oihw,
                         const float* bias, const float* bn_beta, const float* bn_gamma, const float* bn_mean, const float* bn_var, int act, float leaky,
                         float* y);
/* cs_4x_deconv_2s_RGBA.glsl:147-195 line by line for one image: baseCoord = floor((xy + 1) / 2), the four nearest-filtered fetches at
 * baseCoord + {(-1,-1),(0,-1),(-1,0),(0,0)} (zero border: the input sampler clamps to a transparent border, openGLBackend.cpp:41-45),
 * weight indices {0,2,8,10} + x%2 + 4*(y%2) + 16*layer into the per-output-channel vec4 arrays that deconv2dGL.cpp:28-80 builds
 * (lane = input channel within the 4-channel layer, index = 16*layer + tap).  Output 2H x 2W.  No BN / activation (tested through
 * snn_oracle_deconv2d). */
void snn_oracle_deconv4x4s2_shader(const float* x, int H, int W, int IC, int OC, const float* w_oihw, const float* bias, float* y);

void snn_oracle_srand(uint64_t seed);
float snn_oracle_random_float(float a, float b);

#ifdef __cplusplus
}
#endif
#endif
