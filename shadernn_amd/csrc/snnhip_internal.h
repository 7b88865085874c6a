// snnhip_internal.h -- shared declarations of the HIP operator library behind include/snnhip.h.
// gfx950 (MI355X) only: wave64, 160 KiB LDS/CU, 256 CUs in 8 XCDs.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/snnhip.h"

namespace snnhip {

void set_error(const char* fmt, ...);

#define SNNHIP_CHECK_HIP(expr)                                                                         \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            ::snnhip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SNNHIP_E_HIP;                                                                       \
        }                                                                                              \
    } while (0)

#define SNNHIP_REQUIRE(cond, ...)              \
    do {                                       \
        if (!(cond)) {                         \
            ::snnhip::set_error(__VA_ARGS__);  \
            return SNNHIP_E_INVALID;           \
        }                                      \
    } while (0)

// ---- launch trace (snnhip_trace_begin / _end / _report, include/snnhip.h): every kernel launch of the library goes through SNNHIP_LAUNCH; while a
// trace is on it is issued with hipExtLaunchKernelGGL and a fresh event pair -- the dispatch packet's own start / end stamps, what rocprofv3's
// kernel trace reports -- and recorded under the innermost open scope (the plan that launched it and that plan's algorithmic cost).
bool trace_active();
int trace_events(const void* fn, hipStream_t stream, hipEvent_t* start, hipEvent_t* stop);
struct TraceScope {
    int prev = -2; // -2 = inactive
    TraceScope(const std::string& desc, double flops, double bytes);
    explicit TraceScope(const ::snnhip_plan* plan);
    ~TraceScope();
    TraceScope(const TraceScope&) = delete;
    TraceScope& operator=(const TraceScope&) = delete;
};

#define SNNHIP_LAUNCH(kernel, grid, block, lds, stream, ...)                                                                  \
    do {                                                                                                                      \
        if (::snnhip::trace_active()) {                                                                                       \
            hipEvent_t trS_ = nullptr, trE_ = nullptr;                                                                        \
            (void) ::snnhip::trace_events(reinterpret_cast<const void*>(kernel), stream, &trS_, &trE_);                       \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, trS_, trE_, 0, __VA_ARGS__);                              \
        } else {                                                                                                              \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                                \
        }                                                                                                                     \
    } while (0)

// the same for a launch site that may already hold a plan-profile event pair (evS / evE null = none)
#define SNNHIP_LAUNCH_EV(kernel, grid, block, lds, stream, evS, evE, ...)                                                     \
    do {                                                                                                                      \
        hipEvent_t trS_ = evS, trE_ = evE;                                                                                    \
        if (!trS_ && ::snnhip::trace_active()) (void) ::snnhip::trace_events(reinterpret_cast<const void*>(kernel), stream, &trS_, &trE_); \
        hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, trS_, trE_, 0, __VA_ARGS__);                                  \
    } while (0)

// ---- device allocations of the library (tensors, packed weights, scratch).  Plain hipMalloc / hipFree, or -- SNNHIP_GUARD=1 -- the guarded form
// of include/snnhip.h (red zones around every allocation, everything poisoned with 0xFF); capi.hip.
hipError_t dev_malloc_bytes(void** p, size_t bytes, const char* what);
hipError_t dev_free(void* p);
template <class T>
inline hipError_t dev_malloc(T** p, size_t bytes, const char* what = "plan buffer") {
    void* v = nullptr;
    const hipError_t e = dev_malloc_bytes(&v, bytes, what);
    *p = static_cast<T*>(v);
    return e;
}

inline int up_div(int x, int y) { return (x + y - 1) / y; }
inline int round_up(int x, int y) { return up_div(x, y) * y; }

// Conv2DLayer::getOutputScaleDimAdjustment + GenericModelLayer::getOutputDims restated (reference
// core/src/ic2/conv2d.cpp:102-113, genericlayer.cpp:64-90): float arithmetic, truncating conversion.
inline int conv_out_dim(int in, int kernel, int stride, int padA, int padB) {
    float scale = 1 / static_cast<float>(stride);
    float translation = (kernel % 2 != 0)
                            ? 1 + (static_cast<float>(static_cast<unsigned>(padA + padB)) - static_cast<float>(kernel)) / static_cast<float>(stride)
                            : 1 + (static_cast<float>(static_cast<unsigned>(padA + padB - 1)) - static_cast<float>(kernel)) / static_cast<float>(stride);
    float s = scale * static_cast<float>(in);
    if (s < 0.0f) s = 0.0f;
    float t = translation < 0.0f ? 0.0f : translation;
    return static_cast<int>(static_cast<unsigned>(s + t));
}

} // namespace snnhip

struct snnhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr; // where plans launch: the main stream, or the side stream between snnhip_ctx_fork and snnhip_ctx_main
    bool ownsStream = false;
    hipDeviceProp_t props;
    hipStream_t mainStream = nullptr, sideStream = nullptr;
    hipEvent_t forkEvent = nullptr, joinEvent = nullptr;
    void* ksGroup = nullptr; // an open launch group (snnhip_ctx_group_begin; conv2d_ksplit.hip owns the object)
};

struct snnhip_tensor {
    snnhip_ctx* ctx = nullptr;
    float* data = nullptr;
    bool owns = false;
    int n = 0, h = 0, w = 0, c = 0;
    int dtype = SNNHIP_F32; // SNNHIP_F16: `data` points at halfs
    size_t count() const { return static_cast<size_t>(n) * h * w * c; }
    size_t elemSize() const { return dtype == SNNHIP_F16 ? 2 : dtype == SNNHIP_U8 ? 1 : 4; }
    size_t bytes() const { return count() * elemSize(); }
};

struct snnhip_graph {
    snnhip_ctx* ctx = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int nodes = 0;
};

struct snnhip_timer {
    snnhip_ctx* ctx = nullptr;
    hipEvent_t start = nullptr, stop = nullptr;
};

// One executable operator: device-resident parameters + a launch recipe (the reference's InferencePass +
// RenderPass rolled into one object: core/src/ic2/inferencepass.h:31-62, vulkanRenderpass.cpp:103-260).
struct snnhip_plan {
    snnhip_ctx* ctx = nullptr;
    int inDims[4] = {0, 0, 0, 0};   // N,H,W,C of input 0
    int outDims[4] = {0, 0, 0, 0};
    int numInputs = 1;
    int dtype = SNNHIP_F32;   // element type of the tensors this plan runs on ...
    bool anyDtype = false;    // ... unless it adapts to the tensors of each call (element-wise / pooling / shape operators)
    bool u8Input = false;     // snnhip_image_u8_plan_create: the only plan that reads SNNHIP_U8 tensors
    std::string desc;
    double flops = 0, bytes = 0; // algorithmic cost in SURVEY 8(d)'s accounting (a fused plan: the sum over the layers it replaces)
    // what THIS plan's own launches have to move through HBM (inputs once + outputs once + weights): differs from `bytes` only for fused plans,
    // which must set it (the launch trace and bench.py's per-kernel roofline use it); < 0 = same as `bytes`
    double kernelBytes = -1.0;
    double ownBytes() const { return kernelBytes >= 0.0 ? kernelBytes : bytes; }
    std::vector<void*> deviceAllocs; // freed in the destructor

    // per-launch profiling (hipEvent pairs on ctx->stream), see snnhip_plan_profile_enable
    bool profiling = false;
    struct EventPair {
        hipEvent_t start, stop;
    };
    std::vector<std::vector<EventPair>> stepEvents; // [step][launch]
    std::vector<size_t> stepUsed;                   // pairs recorded since the last read

    virtual ~snnhip_plan() {
        for (void* p : deviceAllocs) (void) ::snnhip::dev_free(p);
        for (auto& v : stepEvents)
            for (auto& e : v) {
                (void) hipEventDestroy(e.start);
                (void) hipEventDestroy(e.stop);
            }
    }
    virtual int run(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out) = 0;
    // run() under a launch-trace scope of this plan (snnhip_trace_begin): what the C-ABI and every composite plan call
    int invoke(const snnhip_tensor* const* in, int nIn, snnhip_tensor* out);
    virtual int numSteps() const { return 1; }
    virtual std::string stepDesc(int) const { return desc; }
    virtual void stepCost(int, double* f, double* b) const {
        *f = flops;
        *b = bytes;
    }
    // true when run() brackets its own launches (chains); otherwise snnhip_plan_run_n brackets the single launch
    virtual bool profilesItself() const { return false; }
    int profBegin(int step); // records the start event of a fresh pair
    int profEnd(int step);
    // For kernels this library launches itself: hand out a fresh pair to hipExtLaunchKernelGGL, which stamps the
    // dispatch packet's own start/end instead of enqueueing two marker packets (~3.5 us each on this runtime).
    int profAcquire(int step, hipEvent_t* start, hipEvent_t* stop);

    // uploads host floats into a fresh device buffer owned by the plan
    int upload(const float* host, size_t count, float** dev);
};

namespace snnhip {

// Per-output-channel epilogue parameters {bias, bnScale, bnMean, bnBeta}; bnScale = gamma / max(sqrt(var+1e-3), 1e-4)
// (reference shadertemplate_vk_conv2d.comp:277-288).  Padded with zeros to `padTo` channels.
std::vector<float> make_epilogue_table(int OC, int padTo, int useBias, const float* bias, int useBN, const float* beta,
                                       const float* gamma, const float* mean, const float* var);

// Options registry: the one place a switch is read.  A value set through snnhip_set_option() (C-ABI) wins; otherwise the environment variable of
// the same name, looked up at the call (tests flip switches between plan creations).  DESIGN.md section 8 lists every name.
const char* option(const char* name);

struct ConvGeom {
    int N, H, W, IC, OC, kh, kw, sh, sw;
    int padx, pady; // already resolved: 1x1 => 0 (vk_conv2d_1x1.comp:74-75); else padT / padL (conv2dVulkan.cpp:183-184)
    int padMode, act, useBN;
    float leaky;
    int OH, OW;
    int dtype; // SNNHIP_F32 | SNNHIP_F16
    // fused Pad layer in front of the convolution (chain rule D): H, W above are the PADDED dims the convolution sees, the input tensor is
    // srcH x srcW and padded pixel (y, x) reads source (y - preY, x - preX) resolved with preMode (SNNHIP_PAD_CONSTANT / REPLICATE / REFLECT)
    int preMode = 0, preX = 0, preY = 0, srcH = 0, srcW = 0;
    int preShift = 0; // 1: a nearest x2 UpSampling2D sits in front of the (optional) Pad: the pad resolves against 2*srcH x 2*srcW, then y, x >>= 1
    // fused residual Add behind the convolution (chain rule E): y = addAct(conv(x) + residual); -1 = none
    int addAct = -1;
    float addLeaky = 0.0f;
    // InstanceNorm in front of the convolution [and of its fused Pad / UpSampling] (graph rule I): the kernel applies
    // normAct(x * mul[n][c] + shift[n][c]) (one fp32 fma, rounded to the tensor type: the arithmetic and the rounding point of the norm's own
    // normalise sweep) to every value it stages; zero padding stays zero.  Device pointers owned by the InstanceNorm plan; null = none.
    const float* normShift = nullptr;
    const float* normMul = nullptr;
    int normAct = 0; // none / relu / relu6 / leakyRelu only
    float normLeaky = 0.0f;
};
int resolve_conv_geom(const snnhip_conv2d_desc* d, bool depthwise, ConvGeom* g);

// factories implemented in the .hip translation units
int make_conv2d_generic_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out);
int make_conv2d_mfma_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out);
int make_conv1x1_stream_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out); // tried first by make_conv2d_mfma_plan
int make_conv2d_wino_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out); // tried first by make_conv2d_mfma_plan for fp32 3x3 s1
int make_conv2d_ksplit_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out); // fp32 3x3 s2 (K split over the waves of a block, no reduce launch); tried by make_conv2d_mfma_plan
// conv2d_ksplit.hip: launch groups -- K-split plans run inside a group are launched by its end, two compatible ones as ONE grid
bool ksplit_plan(const snnhip_plan* plan);
int ksplit_group_begin(snnhip_ctx* ctx);
int ksplit_group_end(snnhip_ctx* ctx);
int make_conv2d_stem_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out); // fp16 9x9 s1, IC <= 4; tried first by make_conv2d_mfma_plan
int make_conv2d_stem32_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out); // fp32 stems (IC <= 4), tried first by make_conv2d_mfma_plan
int make_conv2d_wide_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out); // fp16 3x3 s1, large maps; tried first by make_conv2d_mfma_plan
int make_conv2d_widep_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out); // ... IC = OC = 128: persistent blocks; tried first by make_conv2d_wide_plan
int make_conv2d_upconv_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out); // fp16 upsample x2 -> reflect pad 1 -> 3x3 on the low-resolution tensor; tried by make_conv2d_mfma_plan
int make_conv2d_s2march_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out); // fp16 3x3 s2, IC 32 / 64, large maps; tried by make_conv2d_mfma_plan
int make_conv2d_rowmarch_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out); // ... OC <= 4: row-marching strips
int make_conv2d_rowfold_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out); // fp16 k x k, k * OC <= 32
int make_conv2d_thin_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_oihw, const std::vector<float>& epi4, snnhip_plan** out);
int make_depthwise_plan(snnhip_ctx* ctx, const ConvGeom& g, const float* w_chw, const std::vector<float>& epi4, snnhip_plan** out);
int make_dense_plan(snnhip_ctx* ctx, const snnhip_dense_desc& d, const float* w_flat, const float* bias, snnhip_plan** out);
int make_subpixel_plan(snnhip_ctx* ctx, const snnhip_subpixel_desc& d, snnhip_plan** out);
int make_chain_plan(snnhip_ctx* ctx, snnhip_plan* const* plans, int n, snnhip_plan** out);
bool chain_adopt_plan(snnhip_plan* chain, snnhip_plan* p); // the chain deletes p with itself; false if `chain` is not a ChainPlan
// irb_fused.hip (chain rule G): Conv2D 1x1 -> DepthwiseConv2D 3x3 -> Conv2D 1x1 [-> Add with the block input] as one kernel; the plans are only read
int make_irb_plan(snnhip_ctx* ctx, snnhip_plan* expandPlan, snnhip_plan* dwPlan, snnhip_plan* projectPlan, snnhip_plan* addPlan, snnhip_plan** out,
                  snnhip_plan* stemPlan = nullptr);
// dwpw_march.hip (chain rule G without an expand layer, large maps): DepthwiseConv2D 3x3 stride 1 -> Conv2D 1x1 as one row-marching streaming kernel; the plans are only read
int make_dwpw_march_plan(snnhip_ctx* ctx, snnhip_plan* dwPlan, snnhip_plan* pwPlan, snnhip_plan** out);
// ... with the network's stem in front: Conv2D 3x3 stride 2 (3 -> 32 channels) -> DepthwiseConv2D 3x3 -> Conv2D 1x1 as one launch, the stem's output only ever in LDS
int make_stem_dwpw_march_plan(snnhip_ctx* ctx, snnhip_plan* stemPlan, snnhip_plan* dwPlan, snnhip_plan* pwPlan, snnhip_plan** out);

// espcn_stream.hip: the whole ESPCN pattern in one launch (rule C of the chain planner); cfg is an opaque blob
constexpr size_t kStreamCfgBytes = 160;
size_t espcn_stream_step_size();
void espcn_stream_configure(void* cfg, int N, int H, int W, int k1, int act1, float leaky1, int act2, float leaky2, int act3, float leaky3,
                            int computeUnits);
void espcn_stream_describe(const void* cfg, char* buf, size_t n);
int espcn_stream_launch(hipStream_t stream, const void* cfg, const float* x, const float* w1s, const float* ep1, const float* wA2, const float* ep2,
                        const float* w3s, const float* ep3, float* y);

// Conv plans keep their host-side description so that chain fusion can re-pack weights.
struct ConvPlanBase : snnhip_plan {
    ConvGeom g;
    std::vector<float> w_oihw; // host copy
    std::vector<float> epi4;   // host copy of the epilogue table, padded to a multiple of 16
    bool depthwise = false;
    // chain rule F (Conv2D -> InstanceNorm): a convolution that can emit per-tile output statistics switches them on here and publishes where
    // they land (statPart[(n*statTilesY + ty)*statTilesX + tx][2][OC], tiles of statTH x statTW output pixels); false = not supported
    float* statPart = nullptr;
    int statTilesX = 0, statTilesY = 0, statTH = 0, statTW = 0;
    virtual bool enableTileStats() { return false; }
    // ... and a kernel that can also FOLD the records itself (the last block of an image to finish merges that image's records and writes the
    // norm's shift / mul: no fold launches behind the convolution) is given the norm's parameters here; false = not supported
    virtual bool enableNormFold(const struct NormFoldTarget&) { return false; }
    // a kernel whose records are per BLOCK (conv2d_upconv, conv2d_s2march) can only be read by its own in-kernel fold: when enableNormFold fails the
    // chain planner switches the statistics off again and the norm keeps its sweep
    virtual bool tileStatsNeedKernelFold() const { return false; }
    virtual void disableTileStats() {}
};
// the InstanceNorm a convolution folds its tile statistics for: y = x * mul[n][c] + shift[n][c], mul = gamma / sqrt(var + eps), shift = beta - mean * mul
struct NormFoldTarget {
    const float* gamma = nullptr;
    const float* beta = nullptr;
    float* shift = nullptr;
    float* mul = nullptr;
    float eps = 0.0f;
};
bool instancenorm_fold_target(snnhip_plan* normPlan, NormFoldTarget* t);
// eltwise_pool.hip: identify an InstanceNorm plan / run its fold + normalise passes in place from a convolution's tile statistics
bool instancenorm_plan_desc(const snnhip_plan* plan, snnhip_instancenorm_desc* d);
// eltwise_pool.hip: identify a Pooling plan (resolved output dims included)
bool pool2d_plan_desc(const snnhip_plan* plan, snnhip_pool2d_desc* d);
// conv2d_stem_f32.hip, chain rule J: Conv2D 7x7 stride 2 of an RGB image -> MaxPooling2D 3x3 stride 2 (the head of ResNet-18) as one launch: the pooling
// runs in the stem's epilogue (borrows both plans; SNNHIP_E_UNSUPPORTED for anything else)
int make_conv2d_stem32_pool_plan(snnhip_ctx* ctx, snnhip_plan* stemPlan, snnhip_plan* poolPlan, snnhip_plan** out);
// chain rule H: InstanceNorm -> Add(., residual) folded into the norm's normalise sweep (two-input plan; borrows normPlan)
// the Add's output must have the norm's extent (a smaller residual is added top-left aligned, the reference's ragged-Add rule)
int make_instancenorm_add_plan(snnhip_ctx* ctx, snnhip_plan* normPlan, snnhip_plan* addPlan, bool normIsFirstInput, snnhip_plan** out);
// graph rule I (InstanceNorm -> [Pad] -> Conv2D, the normalisation applied while the convolution stages its input)
bool instancenorm_stat_pointers(const snnhip_plan* plan, const float** shift, const float** mul);
// chain rule F: where a convolution left the {mean, M2} records of its output tiles (ConvPlanBase::statPart and its tile grid); part == null = none
struct TileStatsRef {
    const float* part = nullptr;
    int tilesX = 0, tilesY = 0, TH = 0, TW = 0;
    bool folded = false; // the convolution folded the records itself (ConvPlanBase::enableNormFold): shift / mul are ready when it has run
};
// the norm's shift / mul from a statistics sweep over x, or (tiles && tiles->part) from a fold over the producing convolution's tile records
int instancenorm_run_stats(snnhip_plan* plan, const snnhip_tensor* x, const TileStatsRef* tiles = nullptr);
int instancenorm_reserve_tile_stats(snnhip_plan* inPlan, int tilesX, int tilesY);
int instancenorm_apply_tile_stats(snnhip_plan* inPlan, const TileStatsRef& tiles, snnhip_tensor* xy);
// rule F behind rule H: an InstanceNorm -> Add plan (make_instancenorm_add_plan) takes its statistics from tile records from now on; returns its
// norm plan (for instancenorm_reserve_tile_stats), null if `plan` is not such a plan
snnhip_plan* instancenorm_add_use_tile_stats(snnhip_plan* plan, const TileStatsRef& tiles);
struct EltwisePlanBase : snnhip_plan {
    snnhip_eltwise_desc d;
    int mode = 0; // 0 add, 1 activation, 2 batch-norm
};
struct UpsamplePlanBase : snnhip_plan {
    snnhip_upsample_desc d;
    int OH = 0, OW = 0;
};
struct PadPlanBase : snnhip_plan {
    snnhip_pad_desc d;
    int OH = 0, OW = 0;
};
struct SubpixelPlanBase : snnhip_plan {
    snnhip_subpixel_desc d;
};

} // namespace snnhip
