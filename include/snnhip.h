/*
 * snnhip.h -- thin C-ABI between ShaderNN's C++ host side and the MI355X (gfx950) HIP operator kernels.
 *
 * This is the drop-in boundary: plain pointers and sizes only, no C++ or
 * torch types.  Everything the reference's per-layer "render pass" does on the GPU is reachable through these
 * entry points:
 *
 *   reference interface being replaced                                   | entry point here
 *   ---------------------------------------------------------------------+------------------------------------------
 *   createDefaultContext / VulkanBackend ctor                            | snnhip_ctx_create / _create_on_stream
 *     core/src/contextFactory.cpp:21-32, core/src/ic2/vulkanBackend.cpp:28-41
 *   ImageTextureVulkan::resetTexture / upload / download / attach        | snnhip_tensor_alloc / _upload* / _download* / _wrap
 *     core/src/imageTextureVulkan.cpp:188-243, core/inc/snn/imageTexture.h:69-147,301-313
 *   Conv2DLayerVulkan::createCS + VulkanRenderPass ctor (weights, spec    | snnhip_conv2d_plan_create
 *     constants, pipeline)  core/src/ic2/conv2dVulkan.cpp:38-239, vulkanRenderpass.cpp:103-178
 *   SeparableConv2DLayerVulkan::createCS                                 | snnhip_depthwise_plan_create
 *     core/src/ic2/separableconvolutionVulkan.cpp:32-160
 *   DenseLayerVulkan::createCS / DenseLayer::computeImageTexture         | snnhip_dense_plan_create
 *     core/src/ic2/denselayerVulkan.cpp:33-126, denselayer.cpp:27-38
 *   SubpixelLayerVulkan::createCS                                        | snnhip_subpixel_plan_create
 *     core/src/ic2/subpixelmergeVulkan.cpp:29-91
 *   Add/Activation/BatchNorm/Pooling/Pad/Upsample/InstanceNorm/Concat    | snnhip_eltwise_plan_create ... (section "next ops")
 *   VulkanRenderPass::run (bind + Dispatch + barrier)                    | snnhip_plan_run
 *     core/src/ic2/vulkanRenderpass.cpp:181-260
 *   VulkanBackend::sync (QueueSubmitAndWait)                             | snnhip_sync
 *     core/src/ic2/vulkanBackend.cpp:97-106
 *   DeviceTimer (timestamp query pool)  core/inc/snn/deviceTimer.h:20-51 | snnhip_timer_*
 *
 * Conventions: every function returns 0 (SNNHIP_OK) or a negative SNNHIP_E_* code; snnhip_last_error() gives a
 * thread-local human-readable message.  The C++ wrapper turns non-zero into SNN_RIP to keep the reference's
 * abort-on-error behaviour (core/inc/snn/utils.h:57-62).  Tensors are plain NHWC fp32 buffers in HBM (the
 * reference's RGBA "C4HW4" 3-D textures exist only at the API edge: *_c4hw4 upload/download convert).
 * All work is enqueued on the context's HIP stream; nothing synchronises except snnhip_sync, the download
 * functions and snnhip_timer_elapsed_ms.
 */
#ifndef SNNHIP_H
#define SNNHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNNHIP_OK 0
#define SNNHIP_E_INVALID (-1)     /* bad argument / shape mismatch                          */
#define SNNHIP_E_HIP (-2)         /* a HIP runtime call failed (message has the hipError)   */
#define SNNHIP_E_UNSUPPORTED (-3) /* valid request that no kernel variant implements        */
#define SNNHIP_E_NOMEM (-4)
#define SNNHIP_E_GUARD (-5)       /* SNNHIP_GUARD=1: a kernel wrote outside one of the library's device allocations */

typedef struct snnhip_ctx snnhip_ctx;
typedef struct snnhip_tensor snnhip_tensor;
typedef struct snnhip_plan snnhip_plan;
typedef struct snnhip_timer snnhip_timer;

/* activation ids: specialisation constant 15 of the conv shaders (conv2dVulkan.cpp:58-72) */
enum {
    SNNHIP_ACT_NONE = 0,
    SNNHIP_ACT_RELU = 1,
    SNNHIP_ACT_RELU6 = 2,
    SNNHIP_ACT_TANH = 3,
    SNNHIP_ACT_SIGMOID = 4,
    SNNHIP_ACT_LEAKY = 5,
    SNNHIP_ACT_SILU = 6,       /* x*sigmoid(x) */
    SNNHIP_ACT_SILU_QUIRK = 7  /* bit-for-bit model of the reference SiLU bug (vk_conv2d.comp:336-339) */
};
/* padding mode: specialisation constant 16 (conv2dVulkan.cpp:74-81) */
enum { SNNHIP_PAD_NONE = 0, SNNHIP_PAD_CONSTANT = 1, SNNHIP_PAD_REPLICATE = 2, SNNHIP_PAD_REFLECT = 3 };
/* element types.  SNNHIP_F16 = IEEE half storage (the reference's RGBA16F textures / "preferHp", inferencegraph.h ColorFormat::RGBA16F):
 * tensors hold halfs in HBM, kernels convert on load, accumulate in fp32 (fp16-input MFMA for the convolutions) and round to nearest even
 * on store.  The host-side upload / download entry points always speak fp32 and convert. */
enum { SNNHIP_F32 = 0, SNNHIP_F16 = 1, SNNHIP_U8 = 2 /* 8-bit image input of snnhip_image_u8_plan_create only */ };

/* ---- context -------------------------------------------------------------------------------------- */

typedef struct {
    char name[128];
    int compute_units;
    int lds_bytes_per_cu;
    size_t hbm_bytes;
    int device;
} snnhip_device_info;

int snnhip_ctx_create(int device, snnhip_ctx** out);
/* Uses a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) instead of creating one. */
int snnhip_ctx_create_on_stream(int device, void* hip_stream, snnhip_ctx** out);
int snnhip_ctx_destroy(snnhip_ctx* ctx);
int snnhip_ctx_info(snnhip_ctx* ctx, snnhip_device_info* info);
void* snnhip_ctx_stream(snnhip_ctx* ctx);
/* Two independent operators side by side (MI355X-side design; the reference records one command buffer on one queue,
 * vulkanBackend.cpp:80-95).  snnhip_ctx_fork: the context's SIDE stream (created on first use) waits for everything enqueued on the main stream
 * so far, and every plan run on this context goes to the side stream until snnhip_ctx_main switches back (the side work keeps running);
 * snnhip_ctx_join makes the main stream wait for the side stream's work.  Valid inside a graph capture (the side stream joins the
 * capture through the fork event and must be joined before the capture ends).  Used by the host mirror for residual-block pairs such as
 * ResNet's 3x3 stride-2 convolution and the 1x1 stride-2 downsample that read the same tensor. */
int snnhip_ctx_fork(snnhip_ctx* ctx);
int snnhip_ctx_main(snnhip_ctx* ctx);
int snnhip_ctx_join(snnhip_ctx* ctx);
/* Launch groups (HIP extension, round 6): plans that report snnhip_plan_groupable (today: the K-split convolutions of conv2d_ksplit.hip) and are run
 * between _group_begin and _group_end are launched by _group_end -- two independent ones that fit one grid as ONE kernel launch (a ResNet stage entry's
 * 3x3 stride-2 convolution and the 1x1 stride-2 downsample beside it: the short blocks of the second fill the CUs the first leaves idle), anything
 * else one by one.  The caller guarantees that the grouped plans do not consume each other's outputs; every other plan run inside a group is launched
 * at once, as usual.  Valid inside a graph capture. */
int snnhip_ctx_group_begin(snnhip_ctx* ctx);
int snnhip_ctx_group_end(snnhip_ctx* ctx);
int snnhip_plan_groupable(const snnhip_plan* plan);
int snnhip_sync(snnhip_ctx* ctx);
const char* snnhip_last_error(void);
const char* snnhip_version(void);
/* Options: the kernel-selection / fusion switches of DESIGN.md section 8 (names "SNNHIP_..."), process-wide.  A value set here wins over the
 * environment variable of the same name (the fallback, read when a plan is created); value == NULL removes the override.  The reference has no
 * counterpart: its shader variants are picked by MixedInferenceCore's options struct (core/inc/snn/core.h: dumpOutputs, mrtMode, weightMode ...). */
int snnhip_set_option(const char* name, const char* value);
const char* snnhip_get_option(const char* name); /* the effective value (override, else environment), or NULL */
/* Device-side guard mode (SURVEY section 5's "compute-sanitizer equivalent"; the reference has none, core/CMakeLists.txt:235).  With SNNHIP_GUARD=1 in the
 * environment when the library makes its first allocation, every device allocation of the library -- tensors, packed weights, split-K / statistics
 * scratch -- sits between two 64 KiB red zones filled with 0xFF bytes (a NaN in fp32 and in fp16: an out-of-bounds READ that reaches a result
 * poisons it, and fresh tensors / scratch are poisoned the same way, so a read of something never written shows too), the tensor flush against the
 * zone behind it.  snnhip_sync (and snnhip_guard_check) then verifies every red zone of the context's device and returns SNNHIP_E_GUARD naming
 * the allocation a kernel wrote outside of.  snnhip_guard_selftest writes one byte `offset` bytes past the END of a tensor (offset < 0: in front
 * of it) from a kernel -- the deliberate fault the checker's own test uses.  tools/sanitize.sh runs the fuzz and bench-size tests under it. */
int snnhip_guard_active(void);
int snnhip_guard_check(snnhip_ctx* ctx);
int snnhip_guard_selftest(snnhip_ctx* ctx, snnhip_tensor* t, long offset);

/* ---- tensors: NHWC fp32 in HBM ------------------------------------------------------------------- */

int snnhip_tensor_alloc(snnhip_ctx* ctx, int n, int h, int w, int c, int dtype, snnhip_tensor** out);
/* Borrow device memory owned by someone else (a torch tensor, another tensor's storage). Never freed here. */
int snnhip_tensor_wrap(snnhip_ctx* ctx, void* device_ptr, int n, int h, int w, int c, int dtype, snnhip_tensor** out);
int snnhip_tensor_free(snnhip_tensor* t);
int snnhip_tensor_dims(const snnhip_tensor* t, int dims_nhwc[4]);
void* snnhip_tensor_data(const snnhip_tensor* t);
size_t snnhip_tensor_bytes(const snnhip_tensor* t);
int snnhip_tensor_dtype(const snnhip_tensor* t);
int snnhip_tensor_upload(snnhip_tensor* t, const float* host_nhwc);          /* H2D, synchronous */
int snnhip_tensor_download(const snnhip_tensor* t, float* host_nhwc);        /* stream sync + D2H */
/* Reference texture layout [ceil(C/4)][H][W][4] per image (shaderUnitTest.cpp:87-129, image.cpp:216-245). */
int snnhip_tensor_upload_c4hw4(snnhip_tensor* t, const float* host_c4hw4);
int snnhip_tensor_download_c4hw4(const snnhip_tensor* t, float* host_c4hw4);
int snnhip_tensor_fill(snnhip_tensor* t, float value);

/* ---- plans ---------------------------------------------------------------------------------------- */

/* Mirror of the conv shaders' 20 specialisation constants (conv2dVulkan.cpp:182-203) plus batch. */
typedef struct {
    int N, H, W, IC, OC;
    int kh, kw, sh, sw;
    int padT, padB, padL, padR; /* getPaddingOffset order; like the reference the kernel offsets x by padT and y
                                   by padL (conv2dVulkan.cpp:183-184) */
    int padMode;                /* SNNHIP_PAD_* ; ignored for 1x1 (vk_conv2d_1x1.comp has no padding) */
    int act;                    /* SNNHIP_ACT_* */
    float leaky;
    int useBias, useBN;
    int dtype;                  /* SNNHIP_F32 | SNNHIP_F16: type of the input and output tensors (weights are given in fp32 and converted) */
    int OH, OW;                 /* 0 => derived with the reference's float rule (conv2d.cpp:102-113) */
} snnhip_conv2d_desc;

/* w_oihw [OC][IC][kh][kw]; bias [OC] or NULL; BN arrays [OC] or NULL (required when useBN). Host pointers.
 * A half-precision layer whose own input or output tensor would have 2^31 or more elements (a layer behind a fused UpSampling2D / Pad, whose nominal
 * input never exists) gets a DESCRIPTION-ONLY plan: snnhip_chain_plan_create fuses from it (graph rule D), snnhip_plan_run on it alone returns
 * SNNHIP_E_UNSUPPORTED with a message. */
int snnhip_conv2d_plan_create(snnhip_ctx* ctx, const snnhip_conv2d_desc* desc, const float* w_oihw, const float* bias,
                              const float* bn_beta, const float* bn_gamma, const float* bn_mean, const float* bn_var,
                              snnhip_plan** out);

/* depthwise: IC == OC == channels, w_chw [C][kh][kw]; zero padding by tap clipping (vk_depthwise.comp:77-78) */
int snnhip_depthwise_plan_create(snnhip_ctx* ctx, const snnhip_conv2d_desc* desc, const float* w_chw, const float* bias,
                                 const float* bn_beta, const float* bn_gamma, const float* bn_mean, const float* bn_var,
                                 snnhip_plan** out);

/* dense activations follow the reference CPU path (cpulayer.h:38-42,199-261) */
enum {
    SNNHIP_DENSE_IDENTITY = 0,
    SNNHIP_DENSE_RELU = 1,
    SNNHIP_DENSE_LEAKY = 2,
    SNNHIP_DENSE_SIGMOID = 3,
    SNNHIP_DENSE_SOFTMAX = 4,
    SNNHIP_DENSE_TANH = 5,
    SNNHIP_DENSE_SILU_NOOP = 6
};
typedef struct {
    int batch, in_units, out_units;
    int act;
    float leaky;
    int useBias;
} snnhip_dense_desc;
/* w_flat: the JSON "kernel" array exactly as stored; read as [Out][In] (cpulayer.h:162, vk_dense.comp:69-72).
 * Input tensor: any NHWC tensor with H*W*C == in_units per image (flatten order HWC, cpulayer.h:94-113).
 * Output tensor: [batch][1][1][out_units]. */
int snnhip_dense_plan_create(snnhip_ctx* ctx, const snnhip_dense_desc* desc, const float* w_flat, const float* bias,
                             snnhip_plan** out);

enum { SNNHIP_SUBPIXEL_D2S = 0, SNNHIP_SUBPIXEL_VK_QUIRK = 1 };
typedef struct {
    int N, H, W, C;
    int factor; /* reference hard-codes 2 (subpixelmerge.h:26-33) */
    int mode;   /* SNNHIP_SUBPIXEL_* ; tanh is always applied (vk_subpixel.comp:64-66) */
} snnhip_subpixel_desc;
int snnhip_subpixel_plan_create(snnhip_ctx* ctx, const snnhip_subpixel_desc* desc, snnhip_plan** out);

/* ---- element-wise, pooling and shape operators next to the conv path: what a ResNet-18 /
 * MobileNetV2 / Candy graph needs between its convolutions.  All are HBM-bound NHWC kernels (16-byte channel-contiguous accesses).
 *   reference interface replaced (core/src/ic2, core/data/assets/shaders)         | entry point
 *   AddLayerVulkan::createCS addlayerVulkan.cpp:33-114 + vk_add.comp:41-88        | snnhip_add_plan_create (two inputs: snnhip_plan_run_n)
 *   ActivationLayerVulkan activationVulkan.cpp + vk_activation.comp:41-86         | snnhip_activation_plan_create
 *   BatchNormalizationLayerVulkan batchnormVulkan.cpp + vk_batchnorm.comp:54-104  | snnhip_batchnorm_plan_create
 *   MaxPooling2DLayerVulkan maxpool2dVulkan.cpp:33-130 + vk_maxpool2d.comp:42-74  | snnhip_pool2d_plan_create (type 0)
 *   AveragePooling2DLayerVulkan avgpool2dVulkan.cpp + vk_avgpool2d.comp:42-69     | snnhip_pool2d_plan_create (type 1)
 *   AdaptiveAvgPool2dLayerGl adaptiveavgpool2dGL.cpp (mean over the whole image)  | snnhip_pool2d_plan_create (type 1, kernel = input size)
 *   PadLayerVulkan padlayerVulkan.cpp:33-110 + vk_pad.comp:42-71                  | snnhip_pad_plan_create
 *   UpSampling2DLayerVulkan upsampling2dVulkan.cpp:35-123 + vk_upsampling2d_{nearest,bilinear}.comp | snnhip_upsample_plan_create
 *   InstanceNormLayerVulkan instancenormVulkan.cpp + vk_instancenorm.comp:53-160  | snnhip_instancenorm_plan_create */
typedef struct {
    int N, H, W, C;
    int act;      /* SNNHIP_ACT_* (0..6; the quirk id 7 is conv-only) */
    float leaky;
} snnhip_eltwise_desc;
/* Add: desc = OUTPUT dims = max over the two inputs (genericlayer.cpp:64-90).  Inputs smaller than that are accepted at run time with the
 * reference's semantics: the sum is evaluated over the first input's extent, the second input reads 0 outside its own (vk_add.comp:47-49) */
int snnhip_add_plan_create(snnhip_ctx* ctx, const snnhip_eltwise_desc* desc, snnhip_plan** out);
int snnhip_activation_plan_create(snnhip_ctx* ctx, const snnhip_eltwise_desc* desc, snnhip_plan** out);
/* y = act(gamma / max(sqrt(var + 1e-3), 1e-4) * (x - mean) + beta), per channel (vk_batchnorm.comp:63-68) */
int snnhip_batchnorm_plan_create(snnhip_ctx* ctx, const snnhip_eltwise_desc* desc, const float* beta, const float* gamma, const float* mean,
                                 const float* var, snnhip_plan** out);

#define SNNHIP_POOL_MAX 0
#define SNNHIP_POOL_AVG 1
typedef struct {
    int N, H, W, C;
    int kh, kw, sh, sw;
    int padT, padL; /* the Vulkan layers force both to 0 ("not padding on top left", maxpool2dVulkan.cpp:62-64); kept as parameters */
    int OH, OW;     /* 0 = derive with the reference's float rule: in/stride + max(0, 1 - k/stride | 1 - 1/stride) (maxpool2d.cpp:26-36) */
    int same;       /* padding mode for the OH/OW rule: 0 "valid"/"none"/"0", 1 anything else */
    int type;       /* SNNHIP_POOL_*: window clipped to the image; max starts at -100000.0, avg divides by the clipped count */
} snnhip_pool2d_desc;
int snnhip_pool2d_plan_create(snnhip_ctx* ctx, const snnhip_pool2d_desc* desc, snnhip_plan** out);

typedef struct {
    int N, H, W, C;
    int padT, padB, padL, padR; /* output = (H + padT + padB) x (W + padL + padR) (padlayer.cpp:58-67) */
    int mode;                   /* 0 constant (zeros), 1 replicate, 2 reflect (vk_pad.comp:55-66) */
} snnhip_pad_desc;
/* reference quirk kept: the shader shifts x by padT and y by padL (padlayerVulkan.cpp:81-82 feeds uPad = {offsets[0], offsets[2]}) */
int snnhip_pad_plan_create(snnhip_ctx* ctx, const snnhip_pad_desc* desc, snnhip_plan** out);

#define SNNHIP_UPSAMPLE_NEAREST 0
#define SNNHIP_UPSAMPLE_BILINEAR 1
typedef struct {
    int N, H, W, C;
    float scale; /* output = trunc(in * scale) (upsampling2d.h:41-44); the shaders sample at pos * (1/scale) */
    int mode;    /* SNNHIP_UPSAMPLE_* */
} snnhip_upsample_desc;
int snnhip_upsample_plan_create(snnhip_ctx* ctx, const snnhip_upsample_desc* desc, snnhip_plan** out);

typedef struct {
    int N, H, W, C;
    int act;
    float leaky;
    float eps; /* the Vulkan shader hard-codes 1e-5 and ignores the parsed epsilon (vk_instancenorm.comp:118); pass 1e-5f for parity */
} snnhip_instancenorm_desc;
/* y = act((x - mean_hw) * gamma / sqrt(var_hw + eps) + beta), statistics per image and channel, biased variance (computed in one sweep
 * around a per-channel pivot: same value as the shader's two-pass form up to fp32 rounding) */
int snnhip_instancenorm_plan_create(snnhip_ctx* ctx, const snnhip_instancenorm_desc* desc, const float* beta, const float* gamma, snnhip_plan** out);

/* ---- the remaining graph operators (U-Net / YOLOv3-tiny concat, unary, transposed convolution, the
 * moonwellbox "Calculate" op) and the device-side step either side of a model run (input resize + normalise, u8 image -> tensor).
 *   reference interface replaced                                                              | entry point
 *   ConcatenateLayerVulkan::createCS concatenationVulkan.cpp:31-88 + vk_concat.comp:39-52      | snnhip_concat_plan_create (two inputs)
 *   UnaryLayerVulkan::createCS unaryVulkan.cpp:30-83 + vk_unary.comp:40-90                     | snnhip_unary_plan_create
 *   Conv2DTransposeLayerGl::createCS deconv2dGL.cpp:282-343 + cs_4x_deconv_2s_RGBA.glsl:150-195 | snnhip_deconv2d_plan_create
 *   CalculateLayerGl::createFS calculationGL.cpp:28-57 + fs_calculation.glsl:25-41            | snnhip_calculate_plan_create
 *   ImageTextureVulkan::resize imageTextureVulkan.cpp:137-183 + vk_resize.comp:41-62           | snnhip_resize_plan_create
 *   snn::normalize / norm2rgba32f image.cpp:712-796 (ImageTexture::convertToRGBA32FAndNormalize) | snnhip_image_u8_plan_create */
typedef struct {
    int N, H, W;
    int C0, C1; /* channels of input 0 / input 1 */
    int OC;     /* the layer's "outputPlanes".  The shader concatenates TEXEL PLANES (4-channel groups), not channels: output plane p comes
                   from input 0 while p < ceil(C0/4), else from input 1's plane p - ceil(C0/4) (vk_concat.comp:43-49).  For C0 % 4 == 0 that
                   is the ordinary channel concat; otherwise input 1 starts at channel 4*ceil(C0/4), the gap reads 0, and whatever does
                   not fit into ceil(OC/4) planes is dropped -- reproduced as is. */
} snnhip_concat_desc;
int snnhip_concat_plan_create(snnhip_ctx* ctx, const snnhip_concat_desc* desc, snnhip_plan** out);

/* vk_unary.comp:46-88 (UnaryDesc::opType is never parsed from JSON, unary.h:27-32: models get 0 = copy) */
enum { SNNHIP_UNARY_COPY = 0, SNNHIP_UNARY_FIXED = 1, SNNHIP_UNARY_NEG = 2, SNNHIP_UNARY_RCP = 3, SNNHIP_UNARY_SQUARE = 4, SNNHIP_UNARY_EXP = 5, SNNHIP_UNARY_ABS = 6 };
typedef struct {
    int N, H, W, C;
    int op;      /* SNNHIP_UNARY_* */
    float value; /* the constant of SNNHIP_UNARY_FIXED */
} snnhip_unary_desc;
int snnhip_unary_plan_create(snnhip_ctx* ctx, const snnhip_unary_desc* desc, snnhip_plan** out);

/* Transposed convolution as the reference's compute shader states it for k=4, s=2 (cs_4x_deconv_2s_RGBA.glsl:150-182): a correlation of the
 * kernel with the zero-stuffed input, zero outside the input (the GL sampler clamps to a transparent border, openGLBackend.cpp:41-43):
 *     y[oy][ox][o] = act(BN(bias[o] + sum_{iy,ix,i} x[iy][ix][i] * w[o][i][(k-1-p) - oy + s*iy][(k-1-p) - ox + s*ix]))   (taps inside the kernel)
 * with p = (k - s) / 2 and output s*H x s*W for "same", p = 0 and s*H + k - s for anything else (deconv2dGL.cpp:345-355).  The desc is the
 * conv desc: kh == kw, sh == sw, padT carries p, padMode / padB / padL / padR are ignored, OH / OW = 0 derives the "same" extent.
 * Activations: relu, tanh, sigmoid, leakyRelu (deconv2dGL.cpp:198-207); BN eps 1e-3 (cs_4x_deconv_2s_RGBA.glsl:190).
 * Pinned against the reference arithmetic for k=4, s=2 only -- the 3x3 / 4x4 stride-1 GL shaders carry sampler-state-dependent texel
 * tests (cs_3x_deconv_RGBA.glsl:77-87) that have no counterpart off the GL rasteriser; other (k, s) follow the formula above. */
int snnhip_deconv2d_plan_create(snnhip_ctx* ctx, const snnhip_conv2d_desc* desc, const float* w_oihw, const float* bias, const float* bn_beta,
                                const float* bn_gamma, const float* bn_mean, const float* bn_var, snnhip_plan** out);

/* fs_calculation.glsl:25-41: y[..][c] = x[..][c % 4] / x[..][8] for c % 4 < 3, else 0 (every 4-channel output pass repeats the same value) */
typedef struct {
    int N, H, W, C; /* C >= 9: the divisor is channel 8 (texel plane 2, .r) */
    int OC;
} snnhip_calculate_desc;
int snnhip_calculate_plan_create(snnhip_ctx* ctx, const snnhip_calculate_desc* desc, snnhip_plan** out);

/* Input pre-processing on the device: resample to OH x OW and normalise, y = (sample(x) - means[c % 4]) * norms[c % 4] (vk_resize.comp:48-60).
 * Sample position of output pixel (x, y): ((x + 0.5) * W / OW, (y + 0.5) * H / OH) in input texel space, clamp-to-edge; linear = the Vulkan
 * bilinear filter with exact fp32 weights (hardware uses 8-bit sub-texel weights: up to 2^-9 of the local range away), else nearest. */
typedef struct {
    int N, H, W, C;
    int OH, OW; /* ImageTextureVulkan::resize: round(W / xScale), round(H / yScale) (imageTextureVulkan.cpp:150-151) */
    float means[4], norms[4];
    int linear;
} snnhip_resize_desc;
int snnhip_resize_plan_create(snnhip_ctx* ctx, const snnhip_resize_desc* desc, snnhip_plan** out);

/* 8-bit image -> normalised 4-channel tensor, the device-side form of convertToRGBA32FAndNormalize (image.cpp:712-751):
 *   src_channels 4 (RGBA8): y[c] = (u8[c] - means[c]) * norms[c];   3 (RGB8): same for c < 3, alpha = 1;
 *   1 (R8): y[0] = (u8 - means[0]) * norms[0], y[1..3] = (0 - means[0]) * norms[0].
 * The input of snnhip_plan_run is a tensor of dtype SNNHIP_U8 [N][H][W][src_channels]; the output is [N][H][W][4] fp32 or fp16. */
typedef struct {
    int N, H, W;
    int src_channels;
    float means[4], norms[4];
} snnhip_image_u8_desc;
int snnhip_image_u8_plan_create(snnhip_ctx* ctx, const snnhip_image_u8_desc* desc, snnhip_plan** out);
/* raw byte upload for SNNHIP_U8 tensors (nbytes must equal snnhip_tensor_bytes) */
int snnhip_tensor_upload_raw(snnhip_tensor* t, const void* host, size_t nbytes);
/* index of the largest element of image n of t (first one on ties, like std::max_element in MixedInferenceCore::run, core.cpp:228-234,
 * which reports index + 1 as classifierOutput); stream sync + a 4-byte D2H */
int snnhip_tensor_argmax(const snnhip_tensor* t, int n, int* out_index);

/* Try to replace a linear chain of plans (plan[i+1] consumes only plan[i]'s output) by fused kernels.
 * On success *out runs the whole chain in <= n launches; intermediate tensors that become internal are never
 * materialised.  Returns SNNHIP_E_UNSUPPORTED when no fusion rule matches (callers keep the unfused plans).
 * Implemented rule set: see DESIGN.md section 4 (ESPCN: conv5x5(1->16)+conv3x3(16->16), conv3x3(16->4)+subpixel). */
int snnhip_chain_plan_create(snnhip_ctx* ctx, snnhip_plan* const* plans, int n, snnhip_plan** out);

/* Graph-level fusion: the ONE place where an operator DAG is searched for fusable groups (the host mirror's
 * HipBackend::finalizeStages and the Python GraphRunner both call it; the rules themselves are snnhip_chain_plan_create's).
 * nodes[i] describes operator i in execution order: its per-layer plan (NULL = opaque operator that never fuses: a CPU stage, a
 * multi-pass layer), its producers (node index, or -(k+1) for model input k) and whether its tensor must exist (model output, dump).
 * out[i] says what to run instead: plan == NULL -> node i's work moved into a later node's plan and its tensor is never produced;
 * owned == 0 -> the node's own plan (run it with out[i].inputs: a folded identity producer, e.g. a Flatten in front of a Dense layer, is
 * skipped by re-wiring); owned == 1 -> a new fused plan the caller destroys with
 * snnhip_plan_destroy, to be run with out[i].inputs (it produces node i's tensor).  The per-layer plans must outlive the fused ones.
 * Groups searched: Conv2D -> Add where the Add is the convolution's only consumer (rule E, two-input fused plan), and maximal
 * linear runs (each operator the sole consumer of the previous one) handed to the chain planner (rules A-D, F). */
#define SNNHIP_GRAPH_MAX_INPUTS 4
typedef struct snnhip_graph_node {
    snnhip_plan* plan;
    int n_inputs;
    int inputs[SNNHIP_GRAPH_MAX_INPUTS];
    int keep;
} snnhip_graph_node;
typedef struct snnhip_fused_node {
    snnhip_plan* plan;
    int owned;
    int n_inputs;
    int inputs[SNNHIP_GRAPH_MAX_INPUTS];
} snnhip_fused_node;
int snnhip_graph_fuse(snnhip_ctx* ctx, const snnhip_graph_node* nodes, int n, snnhip_fused_node* out);

int snnhip_plan_run(snnhip_plan* plan, const snnhip_tensor* in, snnhip_tensor* out);
/* multi-input ops (add, concat): inputs[n_in] */
int snnhip_plan_run_n(snnhip_plan* plan, const snnhip_tensor* const* inputs, int n_in, snnhip_tensor* out);
int snnhip_plan_output_dims(const snnhip_plan* plan, int dims_nhwc[4]);
/* Human-readable kernel-variant description ("conv2d_mfma_f32 tile=...") for logs and tests. */
int snnhip_plan_describe(const snnhip_plan* plan, char* buf, size_t buflen);
/* Algorithmic work of one run: flops and true-channel HBM bytes (inputs once + outputs once + weights once). */
int snnhip_plan_cost(const snnhip_plan* plan, double* flops, double* bytes);
int snnhip_plan_destroy(snnhip_plan* plan);

/* A plan executes as one or more kernel launches ("steps": 1 for plain operators, one per fused kernel for chains).
 * With profiling enabled every launch is bracketed by a hipEvent pair on the context stream (the analogue of the
 * reference's per-stage DeviceTimer, core/src/ic2/core.cpp:140-153).  snnhip_plan_profile_read waits for the
 * recorded events, returns the summed duration and launch count of step `step` since the last read, and resets. */
int snnhip_plan_num_steps(const snnhip_plan* plan);
int snnhip_plan_step_describe(const snnhip_plan* plan, int step, char* buf, size_t buflen);
int snnhip_plan_step_cost(const snnhip_plan* plan, int step, double* flops, double* bytes);
int snnhip_plan_profile_enable(snnhip_plan* plan, int enable);
int snnhip_plan_profile_read(snnhip_plan* plan, int step, double* total_ms, int* launches);

/* ---- launch trace: per-KERNEL-FUNCTION timing of everything the library launches (measurement only) -------------------------
 * The reference times per STAGE (DeviceTimer around a render pass, core/src/ic2/core.cpp:140-153,392-404); a fused plan or a split-K
 * convolution is several launches per stage, and a roofline is a statement about ONE kernel.  Between snnhip_trace_begin and
 * snnhip_trace_end every kernel launch of the process's plans is issued with its own event pair (the dispatch packet's start / end
 * stamps: the durations rocprofv3 --kernel-trace reports) and booked on the plan that launched it; plans must be RUN, not replayed
 * from a captured graph, while a trace is on.  snnhip_trace_report writes JSON: launches grouped by kernel function (template
 * instantiations listed inside), each with launch count, summed duration, and the algorithmic flops / HBM bytes of the plan
 * invocations whose longest launch it was (a plan's cost = snnhip_plan_cost's flops and, for a fused plan, what the fused launch
 * itself has to move instead of the per-layer sum).  *needed = bytes the full report takes (call with buf == NULL to size it). */
int snnhip_trace_begin(void);
int snnhip_trace_end(void);
int snnhip_trace_report(char* buf, size_t buflen, size_t* needed);

/* ---- device timers (hipEvent pairs on the context stream) ---------------------------------------- */

int snnhip_timer_create(snnhip_ctx* ctx, snnhip_timer** out);
int snnhip_timer_start(snnhip_timer* t);
int snnhip_timer_stop(snnhip_timer* t);
int snnhip_timer_elapsed_ms(snnhip_timer* t, float* ms); /* waits for the stop event */
int snnhip_timer_destroy(snnhip_timer* t);

/* ---- launch graphs ------------------------------------------------------------------------------------
 * The reference records one command buffer per inference and replays it (vulkanBackend.cpp:80-106: prepareRun records, sync submits).
 * The HIP counterpart is a captured hipGraph: every snnhip_plan_run* between begin and end is recorded instead of executed (plans must
 * be run with the same tensors they will be replayed on; profiling must be off), and snnhip_graph_launch replays the whole sequence with
 * one host call -- what a 65-kernel MobileNetV2 or a 54-kernel Candy inference needs to stop being launch-bound. */
typedef struct snnhip_graph snnhip_graph;
int snnhip_graph_begin_capture(snnhip_ctx* ctx);
int snnhip_graph_end_capture(snnhip_ctx* ctx, snnhip_graph** out);
int snnhip_graph_launch(snnhip_graph* g); /* enqueues on the context stream */
int snnhip_graph_num_nodes(const snnhip_graph* g);
int snnhip_graph_destroy(snnhip_graph* g);

#ifdef __cplusplus
}
#endif
#endif
