// layers.cpp -- layer definitions of the hot path + the layer factory
// (reference core/src/ic2/genericlayer.cpp, conv2d.cpp, conv2dVulkan.cpp, separableconvolution*.cpp, denselayer*.cpp,
//  subpixelmergeVulkan.cpp, layerFactory.cpp).  createCS() packs a host-side recipe; the backend turns it into a HIP plan.
#include <mutex>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <unordered_map>

#include "../../include/snnhip.h"
#include "ic2/genericlayer.h"
#include "ic2/layerFactory.h"

using namespace snn;
using namespace snn::dp;

// ------------------------------------------------------------------------------------------------ GenericModelLayer

GenericModelLayer::~GenericModelLayer() {
    prevLayers.clear();
    nextLayers.clear();
}

void GenericModelLayer::init(DeviceBackend* backend, ImageTextureArray& inputMat, ImageTextureArray& outputMat) {
    SNN_LOGD("Layer initialized: %s", name.c_str());
    backend->initRenderPasses(this, inputMat, outputMat);
}

void GenericModelLayer::run(DeviceBackend*, bool dumpOutputs) { // genericlayer.cpp:39-62
    for (size_t passCount = 0; passCount < renderPasses.size(); ++passCount) {
        auto& renderPass = renderPasses[passCount];
        const bool lastPass = passCount == renderPasses.size() - 1;
        if (dumpOutputs && lastPass) {
            if (!renderPass->debugPassInputs(outputDir())) SNN_LOGE("Error dumping inputs for layer %s", name.c_str());
            if (!renderPass->debugPassWeights(outputDir(), static_cast<int>(passCount))) SNN_LOGE("Error dumping weights for layer %s", name.c_str());
        }
        renderPass->run();
        if (dumpOutputs && lastPass) {
            if (!renderPass->debugPassOutput(outputDir())) SNN_LOGE("Error dumping outputs for layer %s", name.c_str());
        }
    }
}

// genericlayer.cpp:64-90: float arithmetic, max() accumulation (a negative translation is clamped to 0), truncation
void GenericModelLayer::getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const {
    width = height = depth = 0U;
    InferenceGraph::Transform acc;
    acc.isFixed = false;
    acc.scaleWidth = acc.scaleHeight = acc.translateWidth = acc.translateHeight = 0.0f;
    const InferenceGraph::Transform t = getOutputScaleDimAdjustment();
    for (auto& dim : inputDims) {
        if (!t.isFixed) {
            acc.scaleWidth = std::max(acc.scaleWidth, t.scaleWidth * dim.width);
            acc.translateWidth = std::max(acc.translateWidth, t.translateWidth);
            acc.scaleHeight = std::max(acc.scaleHeight, t.scaleHeight * dim.height);
            acc.translateHeight = std::max(acc.translateHeight, t.translateHeight);
            width = static_cast<uint32_t>(acc.scaleWidth + acc.translateWidth);
            height = static_cast<uint32_t>(acc.scaleHeight + acc.translateHeight);
            depth = std::max(depth, dim.channels);
        } else {
            width = t.fixedWidth;
            height = t.fixedHeight;
            depth = std::max(depth, dim.channels);
        }
    }
}

void ShaderLayer::createInferencePasses(const LayerGenOptions& options) { // genericlayer.cpp:92-114 (vulkan/compute/FS choice collapses to HIP)
    InferencePassesSptr ret = createCS(options);
    setLayerExecutionType(InferenceGraph::LayerExecutionType::GPU_HIP);
    SNN_ASSERT(ret);
    passes = ret;
}

// ------------------------------------------------------------------------------------------------ shared helpers

static void paddingOffsets(const std::string& paddingT, const std::string& paddingB, const std::string& paddingL, const std::string& paddingR,
                           uint32_t kernelSize, uint32_t (&offsets)[4]) { // conv2d.cpp:39-74 == separableconvolution.cpp:27-62
    const bool isdigit = std::all_of(paddingT.begin(), paddingT.end(), ::isdigit);
    if (isdigit && !paddingT.empty()) {
        offsets[0] = static_cast<uint32_t>(std::stoul(paddingT));
        offsets[1] = static_cast<uint32_t>(std::stoul(paddingB));
        offsets[2] = static_cast<uint32_t>(std::stoul(paddingL));
        offsets[3] = static_cast<uint32_t>(std::stoul(paddingR));
    } else if (paddingT == "valid" || paddingT == "none") {
        offsets[0] = offsets[1] = offsets[2] = offsets[3] = 0;
    } else if (kernelSize > 1) {
        offsets[0] = offsets[1] = offsets[2] = offsets[3] = std::max(kernelSize / 2, 1u);
        if (kernelSize % 2 == 0) {
            offsets[0] -= 1;
            offsets[2] -= 1;
        }
    } else {
        offsets[0] = offsets[1] = offsets[2] = offsets[3] = 0;
    }
}

static InferenceGraph::Transform convTransform(const uint32_t (&offset)[4], uint32_t kernelSize, uint32_t stride) { // conv2d.cpp:102-113
    InferenceGraph::Transform t = InferenceGraph::Transform::identity();
    const float scale = 1 / static_cast<float>(stride);
    float translation;
    if (kernelSize % 2 != 0) {
        translation = 1 + (static_cast<float>(offset[0] + offset[1]) - static_cast<float>(kernelSize)) / static_cast<float>(stride);
    } else {
        translation = 1 + (static_cast<float>(offset[0] + offset[1] - 1) - static_cast<float>(kernelSize)) / static_cast<float>(stride);
    }
    t.scaleWidth = t.scaleHeight = scale;
    t.translateWidth = t.translateHeight = translation;
    return t;
}

static int activationId(const std::string& a) { // conv2dVulkan.cpp:58-72
    if (a == "relu") return SNNHIP_ACT_RELU;
    if (a == "relu6") return SNNHIP_ACT_RELU6;
    if (a == "tanh") return SNNHIP_ACT_TANH;
    if (a == "sigmoid") return SNNHIP_ACT_SIGMOID;
    if (a == "leakyRelu") return SNNHIP_ACT_LEAKY;
    if (a == "SiLU") return getenv("SNN_SILU_QUIRK") ? SNNHIP_ACT_SILU_QUIRK : SNNHIP_ACT_SILU;
    return SNNHIP_ACT_NONE;
}

static int paddingModeId(const std::string& m) { // conv2dVulkan.cpp:74-81
    if (m == "constant") return SNNHIP_PAD_CONSTANT;
    if (m == "replicate") return SNNHIP_PAD_REPLICATE;
    if (m == "reflect") return SNNHIP_PAD_REFLECT;
    return SNNHIP_PAD_NONE;
}

struct BnArrays {
    std::vector<float> beta, gamma, mean, var;
};
static BnArrays bnArrays(bool use, const std::map<std::string, std::vector<float>>& bn) {
    BnArrays a;
    if (use) {
        a.beta = bn.at("beta");
        a.gamma = bn.at("gamma");
        a.mean = bn.at("movingMean");
        a.var = bn.at("movingVariance");
    }
    return a;
}

// ------------------------------------------------------------------------------------------------ Conv2D

void Conv2DDesc::parse(ModelParser& parser, int layerId) { // conv2d.cpp:22-32
    GenericConvDesc::parse(parser, layerId);
    int oc = 0, ic = 0, k = 0, s = 0;
    parser.getConvolutionLayer(layerId, oc, ic, activation, k, s, biases, weightsCvM, useBatchNormalization, batchNormalization, leakyReluAlpha,
                               paddingT, paddingB, paddingL, paddingR, paddingMode, useMultiInputs);
    numOutputPlanes = static_cast<uint32_t>(oc);
    numInputPlanes = static_cast<uint32_t>(ic);
    kernelSize = static_cast<uint32_t>(k);
    stride = static_cast<uint32_t>(s);
}

void Conv2DLayer::getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const { // conv2d.cpp:34-37
    GenericModelLayer::getOutputDims(width, height, depth);
    depth = _desc.numOutputPlanes;
}

void Conv2DLayer::getPaddingOffset(uint32_t (&offsets)[4]) const {
    paddingOffsets(_desc.paddingT, _desc.paddingB, _desc.paddingL, _desc.paddingR, _desc.kernelSize, offsets);
}

InferenceGraph::Transform Conv2DLayer::getOutputScaleDimAdjustment() const {
    uint32_t offset[4];
    getPaddingOffset(offset);
    return convTransform(offset, _desc.kernelSize, _desc.stride);
}

InferencePassesSptr Conv2DLayerHip::createCS(const LayerGenOptions&) const { // counterpart of conv2dVulkan.cpp:38-239
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    uint32_t ow = 0, oh = 0, od = 0;
    getOutputDims(ow, oh, od);
    uint32_t pad[4];
    getPaddingOffset(pad);
    snnhip_conv2d_desc d = {};
    d.N = static_cast<int>(inputDims[0].batch);
    d.H = static_cast<int>(inputDims[0].height);
    d.W = static_cast<int>(inputDims[0].width);
    d.IC = static_cast<int>(_desc.numInputPlanes);
    d.OC = static_cast<int>(_desc.numOutputPlanes);
    d.kh = d.kw = static_cast<int>(_desc.kernelSize);
    d.sh = d.sw = static_cast<int>(_desc.stride);
    d.padT = static_cast<int>(pad[0]);
    d.padB = static_cast<int>(pad[1]);
    d.padL = static_cast<int>(pad[2]);
    d.padR = static_cast<int>(pad[3]);
    d.padMode = paddingModeId(_desc.paddingMode);
    d.act = activationId(_desc.activation);
    d.leaky = _desc.leakyReluAlpha;
    d.useBias = _desc.biases.empty() ? 0 : 1; // conv2dVulkan.cpp:118-121
    d.useBN = _desc.useBatchNormalization ? 1 : 0;
    d.dtype = _desc.preferHp ? SNNHIP_F16 : SNNHIP_F32; // conv2dVulkan.cpp:221-229 picks the _fp16 shader asset on preferHp
    d.OH = static_cast<int>(oh);
    d.OW = static_cast<int>(ow);
    const int taps = d.kh * d.kw;
    SNN_CHK(_desc.weightsCvM.size() == static_cast<size_t>(d.OC) * d.IC);
    std::vector<float> oihw(static_cast<size_t>(d.OC) * d.IC * taps);
    for (size_t m = 0; m < _desc.weightsCvM.size(); ++m)
        for (int t = 0; t < taps; ++t) oihw[m * taps + t] = _desc.weightsCvM[m].at<float>(t);
    std::vector<float> bias(static_cast<size_t>(d.OC), 0.0f);
    for (size_t i = 0; i < _desc.biases.size() && i < bias.size(); ++i) bias[i] = static_cast<float>(_desc.biases[i]);
    BnArrays bn = bnArrays(_desc.useBatchNormalization, _desc.batchNormalization);
    InferencePass& pass = ret->passes[0];
    pass.source = formatString("Conv2D k=%d s=%d %d->%d act=%s", d.kh, d.sh, d.IC, d.OC, _desc.activation.c_str());
    pass.createPlan = [d, oihw, bias, bn](snnhip_ctx* ctx, snnhip_plan** out) {
        return snnhip_conv2d_plan_create(ctx, &d, oihw.data(), bias.data(), d.useBN ? bn.beta.data() : nullptr, d.useBN ? bn.gamma.data() : nullptr,
                                         d.useBN ? bn.mean.data() : nullptr, d.useBN ? bn.var.data() : nullptr, out);
    };
    return ret;
}

// ------------------------------------------------------------------------------------------------ SeparableConv2D (depthwise)

void SeparableConv2DDesc::parse(ModelParser& parser, int layerId) { // separableconvolution.h:38-42
    GenericConvDesc::parse(parser, layerId);
    int oc = 0, ic = 0, k = 0, s = 0;
    parser.getDepthwiseConvolutionLayer(layerId, oc, ic, activation, k, s, biases, weightsCvM, useBatchNormalization, batchNormalization,
                                        leakyReluAlpha, paddingT, paddingB, paddingL, paddingR);
    numOutputPlanes = static_cast<uint32_t>(oc);
    numInputPlanes = static_cast<uint32_t>(ic);
    kernelSize = static_cast<uint32_t>(k);
    stride = static_cast<uint32_t>(s);
}

void SeparableConv2DLayer::getPaddingOffset(uint32_t (&offsets)[4]) const {
    paddingOffsets(_desc.paddingT, _desc.paddingB, _desc.paddingL, _desc.paddingR, _desc.kernelSize, offsets);
}

InferenceGraph::Transform SeparableConv2DLayer::getOutputScaleDimAdjustment() const { // separableconvolution.cpp:64-75
    uint32_t offset[4];
    getPaddingOffset(offset);
    return convTransform(offset, _desc.kernelSize, _desc.stride);
}

void SeparableConv2DLayer::getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const { // separableconvolution.cpp:77-86 (integer rule)
    uint32_t p[4];
    getPaddingOffset(p);
    for (auto& dim : inputDims) {
        width = (dim.width - _desc.kernelSize + p[0] + p[2]) / _desc.stride + 1;
        height = (dim.height - _desc.kernelSize + p[1] + p[3]) / _desc.stride + 1;
        depth = dim.depth;
        break;
    }
}

InferencePassesSptr SeparableConv2DLayerHip::createCS(const LayerGenOptions&) const { // separableconvolutionVulkan.cpp:32-160
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    uint32_t ow = 0, oh = 0, od = 0;
    GenericModelLayer::getOutputDims(ow, oh, od); // the Vulkan pass uses the GENERIC float rule here (:50), not the integer one above
    uint32_t pad[4];
    getPaddingOffset(pad);
    snnhip_conv2d_desc d = {};
    d.N = static_cast<int>(inputDims[0].batch);
    d.H = static_cast<int>(inputDims[0].height);
    d.W = static_cast<int>(inputDims[0].width);
    d.IC = d.OC = static_cast<int>(_desc.numOutputPlanes);
    d.kh = d.kw = static_cast<int>(_desc.kernelSize);
    d.sh = d.sw = static_cast<int>(_desc.stride);
    d.padT = static_cast<int>(pad[0]);
    d.padB = static_cast<int>(pad[1]);
    d.padL = static_cast<int>(pad[2]);
    d.padR = static_cast<int>(pad[3]);
    d.padMode = SNNHIP_PAD_CONSTANT; // spec id 16 is never set: always zero padding by tap clipping (:111-135)
    d.act = activationId(_desc.activation);
    d.leaky = _desc.leakyReluAlpha;
    d.useBias = 1; // no useBias constant: the bias buffer is always read
    d.useBN = _desc.useBatchNormalization ? 1 : 0;
    d.dtype = _desc.preferHp ? SNNHIP_F16 : SNNHIP_F32; // separableconvolutionVulkan.cpp:142 picks the _fp16 shader on preferHp
    d.OH = static_cast<int>(oh);
    d.OW = static_cast<int>(ow);
    const int taps = d.kh * d.kw;
    SNN_CHK(_desc.weightsCvM.size() == static_cast<size_t>(d.OC));
    std::vector<float> chw(static_cast<size_t>(d.OC) * taps);
    for (size_t m = 0; m < _desc.weightsCvM.size(); ++m)
        for (int t = 0; t < taps; ++t) chw[m * taps + t] = _desc.weightsCvM[m].at<float>(t);
    std::vector<float> bias(static_cast<size_t>(d.OC), 0.0f);
    for (size_t i = 0; i < _desc.biases.size() && i < bias.size(); ++i) bias[i] = static_cast<float>(_desc.biases[i]);
    BnArrays bn = bnArrays(_desc.useBatchNormalization, _desc.batchNormalization);
    InferencePass& pass = ret->passes[0];
    pass.source = formatString("DepthwiseConv2D k=%d s=%d c=%d act=%s", d.kh, d.sh, d.OC, _desc.activation.c_str());
    pass.createPlan = [d, chw, bias, bn](snnhip_ctx* ctx, snnhip_plan** out) {
        return snnhip_depthwise_plan_create(ctx, &d, chw.data(), bias.data(), d.useBN ? bn.beta.data() : nullptr, d.useBN ? bn.gamma.data() : nullptr,
                                            d.useBN ? bn.mean.data() : nullptr, d.useBN ? bn.var.data() : nullptr, out);
    };
    return ret;
}

// ------------------------------------------------------------------------------------------------ Dense

InferenceGraph::Transform DenseLayer::getOutputScaleDimAdjustment() const { // denselayer.cpp:40-49
    InferenceGraph::Transform ret;
    ret.isFixed = true;
    ret.fixedWidth = static_cast<uint32_t>(_desc.biases.size());
    ret.fixedHeight = 1;
    ret.fixedDepth = 1;
    ret.fixedBatch = 1;
    return ret;
}

void DenseLayer::getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const {
    width = static_cast<uint32_t>(_desc.biases.size());
    height = 1;
    depth = 1;
}

// The reference runs Dense on the CPU through Eigen (denselayer.cpp:27-38, cpulayer.h:136-266) after a sync + download.
// Here it is a GPU plan with the CPU path's semantics: flat kernel read as [Out][In], HWC flatten order, CPU activation
// table -- including unordered_map::operator[] turning unknown names ("linear", "relu6") into RELU (cpulayer.h:38-42,200).
InferencePassesSptr DenseLayerHip::createCS(const LayerGenOptions&) const {
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    snnhip_dense_desc d = {};
    d.batch = inputDims.empty() ? 1 : static_cast<int>(inputDims[0].batch);
    d.out_units = static_cast<int>(_desc.biases.size());
    std::vector<float> flat;
    for (auto& row : _desc.weights) flat.insert(flat.end(), row.begin(), row.end()); // CPUCommonUtil::flatten2d
    SNN_CHK(d.out_units > 0 && flat.size() % static_cast<size_t>(d.out_units) == 0);
    d.in_units = static_cast<int>(flat.size() / static_cast<size_t>(d.out_units));
    const std::string& a = _desc.activation;
    if (a == "relu") d.act = SNNHIP_DENSE_RELU;
    else if (a == "leakyRelu") d.act = SNNHIP_DENSE_LEAKY;
    else if (a == "sigmoid") d.act = SNNHIP_DENSE_SIGMOID;
    else if (a == "softmax") d.act = SNNHIP_DENSE_SOFTMAX;
    else if (a == "tanh") d.act = SNNHIP_DENSE_TANH;
    else if (a == "SiLU") d.act = SNNHIP_DENSE_SILU_NOOP;
    else if (a == "identity" || a.empty()) d.act = SNNHIP_DENSE_IDENTITY;
    else {
        SNN_LOGW("Dense activation \"%s\" is not in the reference CPU table: it becomes ReLU there (cpulayer.h:38-42), reproduced", a.c_str());
        d.act = SNNHIP_DENSE_RELU;
    }
    d.leaky = _desc.leakyReluAlpha;
    d.useBias = 1;
    std::vector<float> bias = _desc.biases;
    InferencePass& pass = ret->passes[0];
    pass.source = formatString("Dense %d->%d act=%s", d.in_units, d.out_units, a.c_str());
    pass.createPlan = [d, flat, bias](snnhip_ctx* ctx, snnhip_plan** out) { return snnhip_dense_plan_create(ctx, &d, flat.data(), bias.data(), out); };
    return ret;
}

// ------------------------------------------------------------------------------------------------ Subpixel

InferencePassesSptr SubpixelLayerHip::createCS(const LayerGenOptions&) const { // subpixelmergeVulkan.cpp:29-91
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    snnhip_subpixel_desc d = {};
    d.N = static_cast<int>(inputDims[0].batch);
    d.H = static_cast<int>(inputDims[0].height);
    d.W = static_cast<int>(inputDims[0].width);
    d.C = static_cast<int>(inputDims[0].channels);
    d.factor = 2;
    // default = true depth-to-space (GL shader / Keras); SNN_SUBPIXEL_VK_QUIRK=1 reproduces vk_subpixel.comp:57-66
    d.mode = getenv("SNN_SUBPIXEL_VK_QUIRK") ? SNNHIP_SUBPIXEL_VK_QUIRK : SNNHIP_SUBPIXEL_D2S;
    InferencePass& pass = ret->passes[0];
    pass.source = "Subpixel depth_to_space(2)+tanh";
    pass.createPlan = [d](snnhip_ctx* ctx, snnhip_plan** out) { return snnhip_subpixel_plan_create(ctx, &d, out); };
    return ret;
}

// ------------------------------------------------------------------------------------------------ operators between the convolutions

static snnhip_eltwise_desc eltwiseDesc(const InferenceGraph::IODesc& in, const std::string& activation, float leaky) {
    snnhip_eltwise_desc d = {};
    d.N = static_cast<int>(in.batch);
    d.H = static_cast<int>(in.height);
    d.W = static_cast<int>(in.width);
    d.C = static_cast<int>(in.channels);
    int act = activationId(activation);
    if (act == SNNHIP_ACT_SILU_QUIRK) act = SNNHIP_ACT_SILU; // the 4-pixel quirk exists in the conv shader only
    d.act = act;
    d.leaky = leaky;
    return d;
}

InferencePassesSptr AddLayerHip::createCS(const LayerGenOptions&) const { // addlayerVulkan.cpp:33-114
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    SNN_CHK(inputDims.size() == 2);
    snnhip_eltwise_desc d = eltwiseDesc(inputDims[0], _desc.activation, _desc.leakyReluAlpha);
    uint32_t ow = 0, oh = 0, od = 0;
    GenericModelLayer::getOutputDims(ow, oh, od); // max over the inputs (addlayerVulkan.cpp:48-52); the plan handles smaller inputs
    d.W = static_cast<int>(ow);
    d.H = static_cast<int>(oh);
    ret->passes[0].source = formatString("Add act=%s", _desc.activation.c_str());
    ret->passes[0].createPlan = [d](snnhip_ctx* ctx, snnhip_plan** out) { return snnhip_add_plan_create(ctx, &d, out); };
    return ret;
}

InferencePassesSptr ActivationLayerHip::createCS(const LayerGenOptions&) const { // activationVulkan.cpp
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    const snnhip_eltwise_desc d = eltwiseDesc(inputDims[0], _desc.activation, _desc.leakyReluAlpha);
    ret->passes[0].source = formatString("Activation %s", _desc.activation.c_str());
    ret->passes[0].createPlan = [d](snnhip_ctx* ctx, snnhip_plan** out) { return snnhip_activation_plan_create(ctx, &d, out); };
    return ret;
}

InferencePassesSptr BatchNormalizationLayerHip::createCS(const LayerGenOptions&) const { // batchnormVulkan.cpp
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    const snnhip_eltwise_desc d = eltwiseDesc(inputDims[0], _desc.activation, _desc.leakyReluAlpha);
    const BnArrays bn = bnArrays(true, _desc.batchNormalization);
    SNN_CHK(bn.beta.size() >= static_cast<size_t>(d.C) && bn.var.size() >= static_cast<size_t>(d.C));
    ret->passes[0].source = formatString("BatchNormalization c=%d act=%s", d.C, _desc.activation.c_str());
    ret->passes[0].createPlan = [d, bn](snnhip_ctx* ctx, snnhip_plan** out) {
        return snnhip_batchnorm_plan_create(ctx, &d, bn.beta.data(), bn.gamma.data(), bn.mean.data(), bn.var.data(), out);
    };
    return ret;
}

static InferenceGraph::Transform poolTransform(const std::string& padding, uint32_t kernelSize, uint32_t stride) { // maxpool2d.cpp:26-36 == avgpool2d.cpp:20-29
    InferenceGraph::Transform t = InferenceGraph::Transform::identity();
    const float scale = 1.0f / static_cast<float>(stride);
    float translation;
    if (padding == "0" || padding == "valid" || padding == "none") translation = 1.0f - (static_cast<float>(kernelSize) / static_cast<float>(stride));
    else translation = 1.0f - 1.0f / static_cast<float>(stride);
    t.scaleWidth = t.scaleHeight = scale;
    t.translateWidth = t.translateHeight = translation;
    return t;
}

InferenceGraph::Transform MaxPooling2DLayer::getOutputScaleDimAdjustment() const { return poolTransform(_desc.paddingT, _desc.kernelSize, _desc.stride); }
InferenceGraph::Transform AveragePooling2DLayer::getOutputScaleDimAdjustment() const { return poolTransform(_desc.padding, _desc.kernelSize, _desc.stride); }

static InferencePassesSptr poolPasses(const InferenceGraph::IODesc& in, uint32_t ow, uint32_t oh, int kh, int kw, int sh, int sw, int type, const char* what) {
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    snnhip_pool2d_desc d = {};
    d.N = static_cast<int>(in.batch);
    d.H = static_cast<int>(in.height);
    d.W = static_cast<int>(in.width);
    d.C = static_cast<int>(in.channels);
    d.kh = kh;
    d.kw = kw;
    d.sh = sh;
    d.sw = sw;
    d.padT = d.padL = 0; // "Hack it. Looks like not padding on top left in NCNN" (maxpool2dVulkan.cpp:62-64, avgpool2dVulkan.cpp:58-60)
    d.OH = static_cast<int>(oh);
    d.OW = static_cast<int>(ow);
    d.type = type;
    ret->passes[0].source = formatString("%s k=%dx%d s=%d c=%d", what, kh, kw, sh, d.C);
    ret->passes[0].createPlan = [d](snnhip_ctx* ctx, snnhip_plan** out) { return snnhip_pool2d_plan_create(ctx, &d, out); };
    return ret;
}

InferencePassesSptr MaxPooling2DLayerHip::createCS(const LayerGenOptions&) const { // maxpool2dVulkan.cpp:33-130
    uint32_t ow = 0, oh = 0, od = 0;
    GenericModelLayer::getOutputDims(ow, oh, od);
    const int k = static_cast<int>(_desc.kernelSize), s = static_cast<int>(_desc.stride);
    return poolPasses(inputDims[0], ow, oh, k, k, s, s, SNNHIP_POOL_MAX, "MaxPooling2D");
}

InferencePassesSptr AveragePooling2DLayerHip::createCS(const LayerGenOptions&) const { // avgpool2dVulkan.cpp
    uint32_t ow = 0, oh = 0, od = 0;
    GenericModelLayer::getOutputDims(ow, oh, od);
    const int k = static_cast<int>(_desc.kernelSize), s = static_cast<int>(_desc.stride);
    return poolPasses(inputDims[0], ow, oh, k, k, s, s, SNNHIP_POOL_AVG, "AveragePooling2D");
}

// adaptiveavgpool2dGL.cpp averages the whole input (N_DIMS = width*height taps): target size 1
void AdaptiveAvgPool2dLayer::getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const {
    width = height = 1;
    depth = inputDims.empty() ? _desc.numOutputPlanes : inputDims[0].channels;
}

InferencePassesSptr AdaptiveAvgPool2dLayerHip::createCS(const LayerGenOptions&) const {
    if (_desc.targetSize != 1) SNN_LOGW("AdaptiveAvgPool2d: target size %d requested; like the reference shader this averages the whole image", _desc.targetSize);
    const int h = static_cast<int>(inputDims[0].height), w = static_cast<int>(inputDims[0].width);
    return poolPasses(inputDims[0], 1, 1, h, w, h, w, SNNHIP_POOL_AVG, "AdaptiveAvgPool2d");
}

void FlattenLayer::getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const { // flattenlayer.cpp:58-62
    width = inputDims[0].width * inputDims[0].height * inputDims[0].channels;
    height = 1;
    depth = 1;
}

// The reference flattens on the CPU in HWC order with an optional activation (flattenlayer.cpp:29-46, cpulayer.h:94-113).  NHWC
// memory already is that order, so this is one activation kernel writing into the W*H*C x 1 x 1 output tensor (no download).
InferencePassesSptr FlattenLayerHip::createCS(const LayerGenOptions&) const {
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    std::string act = _desc.activation == "linear" ? std::string() : _desc.activation;
    const snnhip_eltwise_desc d = eltwiseDesc(inputDims[0], act, _desc.leakyReluAlpha);
    ret->passes[0].source = "Flatten (HWC)";
    ret->passes[0].createPlan = [d](snnhip_ctx* ctx, snnhip_plan** out) { return snnhip_activation_plan_create(ctx, &d, out); };
    return ret;
}

void PadLayer::getPaddingOffset(uint32_t (&offsets)[4]) const { // padlayer.cpp:27-56 (no even-kernel correction here)
    const std::string& paddingT = _desc.paddingT;
    const bool isdigit = !paddingT.empty() && std::all_of(paddingT.begin(), paddingT.end(), ::isdigit);
    if (isdigit) {
        offsets[0] = static_cast<uint32_t>(std::stoul(_desc.paddingT));
        offsets[1] = static_cast<uint32_t>(std::stoul(_desc.paddingB));
        offsets[2] = static_cast<uint32_t>(std::stoul(_desc.paddingL));
        offsets[3] = static_cast<uint32_t>(std::stoul(_desc.paddingR));
    } else if (paddingT == "valid" || paddingT == "none" || _desc.kernelSize <= 1) {
        offsets[0] = offsets[1] = offsets[2] = offsets[3] = 0;
    } else {
        offsets[0] = offsets[1] = offsets[2] = offsets[3] = std::max(_desc.kernelSize / 2, 1u);
    }
}

InferenceGraph::Transform PadLayer::getOutputScaleDimAdjustment() const { // padlayer.cpp:58-67
    uint32_t offset[4];
    getPaddingOffset(offset);
    InferenceGraph::Transform t = InferenceGraph::Transform::identity();
    t.translateWidth = static_cast<float>(offset[2] + offset[3]);
    t.translateHeight = static_cast<float>(offset[0] + offset[1]);
    return t;
}

InferencePassesSptr PadLayerHip::createCS(const LayerGenOptions&) const { // padlayerVulkan.cpp:33-110
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    uint32_t p[4];
    getPaddingOffset(p);
    snnhip_pad_desc d = {};
    d.N = static_cast<int>(inputDims[0].batch);
    d.H = static_cast<int>(inputDims[0].height);
    d.W = static_cast<int>(inputDims[0].width);
    d.C = static_cast<int>(inputDims[0].channels);
    d.padT = static_cast<int>(p[0]);
    d.padB = static_cast<int>(p[1]);
    d.padL = static_cast<int>(p[2]);
    d.padR = static_cast<int>(p[3]);
    d.mode = _desc.mode == "replicate" ? 1 : (_desc.mode == "reflect" ? 2 : 0); // padlayerVulkan.cpp:53-60
    ret->passes[0].source = formatString("Pad %s t%d b%d l%d r%d", _desc.mode.c_str(), d.padT, d.padB, d.padL, d.padR);
    ret->passes[0].createPlan = [d](snnhip_ctx* ctx, snnhip_plan** out) { return snnhip_pad_plan_create(ctx, &d, out); };
    return ret;
}

InferencePassesSptr InstanceNormLayerHip::createCS(const LayerGenOptions&) const { // instancenormVulkan.cpp
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    const snnhip_eltwise_desc e = eltwiseDesc(inputDims[0], _desc.activation == "Relu" ? std::string("relu") : _desc.activation, _desc.leakyReluAlpha);
    snnhip_instancenorm_desc d = {};
    d.N = e.N; d.H = e.H; d.W = e.W; d.C = e.C; d.act = e.act; d.leaky = e.leaky;
    d.eps = 1e-5f; // the shader hard-codes 0.00001 and ignores the parsed epsilon (vk_instancenorm.comp:118)
    if (_desc.epsilon != 1e-5f) SNN_LOGW("InstanceNorm: epsilon %g in the model; the reference shader uses 1e-5 regardless, reproduced", _desc.epsilon);
    const std::vector<float> beta = _desc.instanceNormalization.at("beta"), gamma = _desc.instanceNormalization.at("gamma");
    SNN_CHK(beta.size() >= static_cast<size_t>(d.C) && gamma.size() >= static_cast<size_t>(d.C));
    ret->passes[0].source = formatString("InstanceNorm c=%d act=%s", d.C, _desc.activation.c_str());
    ret->passes[0].createPlan = [d, beta, gamma](snnhip_ctx* ctx, snnhip_plan** out) {
        return snnhip_instancenorm_plan_create(ctx, &d, beta.data(), gamma.data(), out);
    };
    return ret;
}

InferencePassesSptr UpSampling2DLayerHip::createCS(const LayerGenOptions&) const { // upsampling2dVulkan.cpp:35-123
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    snnhip_upsample_desc d = {};
    d.N = static_cast<int>(inputDims[0].batch);
    d.H = static_cast<int>(inputDims[0].height);
    d.W = static_cast<int>(inputDims[0].width);
    d.C = static_cast<int>(inputDims[0].channels);
    d.scale = _desc.scale;
    d.mode = _desc.interpolationType == "bilinear" ? SNNHIP_UPSAMPLE_BILINEAR : SNNHIP_UPSAMPLE_NEAREST; // :58-70
    ret->passes[0].source = formatString("UpSampling2D %s x%g", _desc.interpolationType.c_str(), d.scale);
    ret->passes[0].createPlan = [d](snnhip_ctx* ctx, snnhip_plan** out) { return snnhip_upsample_plan_create(ctx, &d, out); };
    return ret;
}

// ------------------------------------------------------------------------------------------------ SURVEY 8f rank 4

InferencePassesSptr ConcatenateLayerHip::createCS(const LayerGenOptions&) const { // concatenationVulkan.cpp:31-88
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    SNN_CHK(inputDims.size() == 2);
    snnhip_concat_desc d = {};
    d.N = static_cast<int>(inputDims[0].batch);
    d.H = static_cast<int>(inputDims[0].height);
    d.W = static_cast<int>(inputDims[0].width);
    d.C0 = static_cast<int>(inputDims[0].channels);
    d.C1 = static_cast<int>(inputDims[1].channels);
    d.OC = static_cast<int>(_desc.numOutputPlanes); // :56
    ret->passes[0].source = formatString("Concatenate %d+%d->%d", d.C0, d.C1, d.OC);
    ret->passes[0].createPlan = [d](snnhip_ctx* ctx, snnhip_plan** out) { return snnhip_concat_plan_create(ctx, &d, out); };
    return ret;
}

InferencePassesSptr UnaryLayerHip::createCS(const LayerGenOptions&) const { // unaryVulkan.cpp:30-83
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    snnhip_unary_desc d = {};
    d.N = static_cast<int>(inputDims[0].batch);
    d.H = static_cast<int>(inputDims[0].height);
    d.W = static_cast<int>(inputDims[0].width);
    d.C = static_cast<int>(inputDims[0].channels);
    d.op = _desc.opType;
    d.value = _desc.opValue;
    ret->passes[0].source = formatString("Unary op=%d", d.op);
    ret->passes[0].createPlan = [d](snnhip_ctx* ctx, snnhip_plan** out) { return snnhip_unary_plan_create(ctx, &d, out); };
    return ret;
}

InferencePassesSptr CalculateLayerHip::createCS(const LayerGenOptions&) const { // calculationGL.cpp:28-57
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    snnhip_calculate_desc d = {};
    d.N = static_cast<int>(inputDims[0].batch);
    d.H = static_cast<int>(inputDims[0].height);
    d.W = static_cast<int>(inputDims[0].width);
    d.C = static_cast<int>(inputDims[0].channels);
    d.OC = static_cast<int>(_desc.numOutputPlanes);
    ret->passes[0].source = formatString("Calculate %d->%d", d.C, d.OC);
    ret->passes[0].createPlan = [d](snnhip_ctx* ctx, snnhip_plan** out) { return snnhip_calculate_plan_create(ctx, &d, out); };
    return ret;
}

InferencePassesSptr Conv2DTransposeLayerHip::createCS(const LayerGenOptions&) const { // deconv2dGL.cpp:282-343
    auto ret = std::make_shared<InferencePasses>();
    ret->passes.resize(1);
    uint32_t ow = 0, oh = 0, od = 0;
    getOutputDims(ow, oh, od);
    snnhip_conv2d_desc d = {};
    d.N = static_cast<int>(inputDims[0].batch);
    d.H = static_cast<int>(inputDims[0].height);
    d.W = static_cast<int>(inputDims[0].width);
    d.IC = static_cast<int>(_desc.numInputPlanes);
    d.OC = static_cast<int>(_desc.numOutputPlanes);
    d.kh = d.kw = static_cast<int>(_desc.kernelSize);
    d.sh = d.sw = static_cast<int>(_desc.stride);
    d.padT = _desc.paddingT == "same" ? static_cast<int>(_desc.kernelSize - _desc.stride) / 2 : 0;
    const std::string& a = _desc.activation; // deconv2dGL.cpp:198-207: anything else leaves s untouched
    d.act = a == "relu" ? SNNHIP_ACT_RELU : a == "tanh" ? SNNHIP_ACT_TANH : a == "sigmoid" ? SNNHIP_ACT_SIGMOID : a == "leakyRelu" ? SNNHIP_ACT_LEAKY : SNNHIP_ACT_NONE;
    d.leaky = _desc.leakyReluAlpha;
    d.useBias = _desc.biases.empty() ? 0 : 1;
    d.useBN = _desc.useBatchNormalization ? 1 : 0;
    d.OH = static_cast<int>(oh);
    d.OW = static_cast<int>(ow);
    const int taps = d.kh * d.kw;
    SNN_CHK(_desc.weightsCvM.size() == static_cast<size_t>(d.OC) * d.IC); // weightMatrices[idxOutput * numInputPlanes + idxInput], deconv2dGL.cpp:93-96
    std::vector<float> oihw(static_cast<size_t>(d.OC) * d.IC * taps);
    for (size_t m = 0; m < _desc.weightsCvM.size(); ++m)
        for (int t = 0; t < taps; ++t) oihw[m * taps + t] = _desc.weightsCvM[m].at<float>(t);
    std::vector<float> bias(static_cast<size_t>(d.OC), 0.0f);
    for (size_t i = 0; i < _desc.biases.size() && i < bias.size(); ++i) bias[i] = static_cast<float>(_desc.biases[i]);
    BnArrays bn = bnArrays(_desc.useBatchNormalization, _desc.batchNormalization);
    ret->passes[0].source = formatString("Conv2DTranspose k=%d s=%d %d->%d act=%s", d.kh, d.sh, d.IC, d.OC, a.c_str());
    ret->passes[0].createPlan = [d, oihw, bias, bn](snnhip_ctx* ctx, snnhip_plan** out) {
        return snnhip_deconv2d_plan_create(ctx, &d, oihw.data(), bias.data(), d.useBN ? bn.beta.data() : nullptr, d.useBN ? bn.gamma.data() : nullptr,
                                           d.useBN ? bn.mean.data() : nullptr, d.useBN ? bn.var.data() : nullptr, out);
    };
    return ret;
}

// ---- YOLO v3 tiny head: decode + NMS on the CPU, as the reference does (yololayer.cpp:27-226)
namespace {
const int32_t kYoloGridScale[] = {32, 16};
const int32_t kYoloGridChannel = 3, kYoloClasses = 1, kYoloFixed = 5, kYoloOutputs = kYoloClasses + kYoloFixed;
const float kYoloAnchors[] = {10.f, 14.f, 23.f, 27.f, 37.f, 58.f, 81.f, 82.f, 135.f, 169.f, 344.f, 319.f};
const int kYoloMasks[] = {3, 4, 5, 1, 2, 3};
struct YoloBox {
    int cls;
    float score, x, y, w, h;
};
float yoloIoU(const YoloBox& a, const YoloBox& b) { // yololayer.cpp:57-72
    const float x0 = std::max(a.x, b.x), y0 = std::max(a.y, b.y);
    const float x1 = std::min(a.x + a.w, b.x + b.w), y1 = std::min(a.y + a.h, b.y + b.h);
    if (x1 < x0 || y1 < y0) return 0.0f;
    const float inter = (x1 - x0) * (y1 - y0);
    return inter / (a.w * a.h + b.w * b.h - inter);
}
inline float sigmoidStd(float x) { return 1.0f / (1.0f + std::exp(-x)); }
} // namespace

std::vector<std::vector<float>> YOLOLayer::decode(const std::vector<const float*>& heads, int netSize) {
    const float confidenceThreshold = 0.35f, iouThreshold = 0.45f; // yololayer.cpp:182-183
    std::vector<YoloBox> boxes;
    for (size_t idx = 0; idx < heads.size() && idx < 2; ++idx) {
        const int grid = netSize / kYoloGridScale[idx];
        const float* data = heads[idx];
        size_t index = 0;
        for (int gy = 0; gy < grid; ++gy)
            for (int gx = 0; gx < grid; ++gx)
                for (int gc = 0; gc < kYoloGridChannel; ++gc) { // yololayer.cpp:118-160
                    int cls = 0;
                    float maxLogit = -FLT_MAX;
                    for (int i = kYoloFixed; i < kYoloOutputs; ++i)
                        if (data[index + i] > maxLogit) {
                            maxLogit = data[index + i];
                            cls = i - kYoloFixed;
                        }
                    const int anchor = kYoloMasks[gc + static_cast<int>(idx) * kYoloGridChannel];
                    const float prob = 1.f / ((1.f + std::exp(-data[index + 4]) * (1.f + std::exp(-maxLogit)))); // :147, as written
                    if (prob > confidenceThreshold) {
                        const float cx = (gx + sigmoidStd(data[index + 0])) / grid, cy = (gy + sigmoidStd(data[index + 1])) / grid;
                        const float w = std::exp(data[index + 2]) * kYoloAnchors[anchor * 2] / static_cast<float>(kYoloGridScale[idx] * grid);
                        const float h = std::exp(data[index + 3]) * kYoloAnchors[anchor * 2 + 1] / static_cast<float>(kYoloGridScale[idx] * grid);
                        boxes.push_back({cls, prob, cx - w / 2, cy - h / 2, w, h});
                    }
                    index += kYoloOutputs; // the tensors here carry the true 18 channels: no 4-channel texture padding to skip (:158)
                }
    }
    std::stable_sort(boxes.begin(), boxes.end(), [](const YoloBox& l, const YoloBox& r) { return l.score > r.score; });
    std::vector<char> merged(boxes.size(), 0);
    std::vector<std::vector<float>> out;
    for (size_t i = 0; i < boxes.size(); ++i) { // Nms, yololayer.cpp:74-112
        if (merged[i]) continue;
        for (size_t j = i + 1; j < boxes.size(); ++j)
            if (!merged[j] && boxes[i].cls == boxes[j].cls && yoloIoU(boxes[i], boxes[j]) > iouThreshold) merged[j] = 1;
        out.push_back({static_cast<float>(boxes[i].cls), boxes[i].score, boxes[i].x, boxes[i].y, boxes[i].w, boxes[i].h});
    }
    return out;
}

void YOLOLayer::computeImageTexture(ImageTextureArray& inputTex, ImageTextureArray& outputTex) { // yololayer.cpp:177-226
    std::vector<std::vector<float>> host(inputTex.size());
    std::vector<const float*> heads;
    for (size_t i = 0; i < inputTex.size(); ++i) {
        host[i].resize(static_cast<size_t>(inputTex[i].width()) * inputTex[i].height() * inputTex[i].channels());
        inputTex[i].downloadNHWC(host[i].data());
        heads.push_back(host[i].data());
    }
    outputTex[0].setOutputMat(decode(heads));
}

// ------------------------------------------------------------------------------------------------ layer factory

static std::unordered_map<std::string, LayerCreator> LayerRegistryDict;

static GenericModelLayer* InputLayerCreator(ModelParser& parser, int i, bool) {
    InputLayerDesc desc;
    desc.parse(parser, i);
    return new InputLayerLayer(desc);
}
#define DEFINE_HIP_CREATOR(layer)                                                   \
    static GenericModelLayer* layer##Creator(ModelParser& parser, int i, bool) {    \
        layer##Desc desc;                                                           \
        desc.parse(parser, i);                                                      \
        return new layer##LayerHip(std::move(desc));                                \
    }                                                                               \
    GenericModelLayer* snn::dp::layer##Creator1(layer##Desc&& desc, bool) { return new layer##LayerHip(std::move(desc)); }
DEFINE_HIP_CREATOR(Conv2D)
DEFINE_HIP_CREATOR(SeparableConv2D)
DEFINE_HIP_CREATOR(Dense)
DEFINE_HIP_CREATOR(Subpixel)
DEFINE_HIP_CREATOR(Add)
DEFINE_HIP_CREATOR(Activation)
DEFINE_HIP_CREATOR(BatchNormalization)
DEFINE_HIP_CREATOR(MaxPooling2D)
DEFINE_HIP_CREATOR(AveragePooling2D)
DEFINE_HIP_CREATOR(AdaptiveAvgPool2d)
DEFINE_HIP_CREATOR(Flatten)
DEFINE_HIP_CREATOR(Pad)
DEFINE_HIP_CREATOR(InstanceNorm)
DEFINE_HIP_CREATOR(UpSampling2D)
DEFINE_HIP_CREATOR(Concatenate)
DEFINE_HIP_CREATOR(Unary)
DEFINE_HIP_CREATOR(Calculate)
DEFINE_HIP_CREATOR(Conv2DTranspose)
static GenericModelLayer* YOLOCreator(ModelParser& parser, int i, bool) {
    YOLODesc desc;
    desc.parse(parser, i);
    return new YOLOLayer(std::move(desc));
}

// The registry is shared by every model of the process and the replica threads of snn_pool_create build their models at the same time: all
// writes happen under registryMutex, the built-in table is filled exactly once (std::call_once), and createLayerInstance copies the creator out
// under the lock before it calls it.  (The reference is single-threaded here, layerFactory.cpp:102-129.)
static std::mutex registryMutex;
static std::once_flag registryOnce;

void snn::dp::registerLayer(const std::string& layerName, LayerCreator creator) {
    std::lock_guard<std::mutex> lock(registryMutex);
    LayerRegistryDict.emplace(layerName, creator);
}

static void fillLayerRegistry();
void snn::dp::initLayerRegisty() { std::call_once(registryOnce, fillLayerRegistry); }

static void fillLayerRegistry() { // layerFactory.cpp:109-129 (the hot-path operators + the element-wise / pooling / shape operators around them)
    using snn::dp::registerLayer;
    registerLayer("InputLayer", InputLayerCreator);
    registerLayer("Conv2D", Conv2DCreator);
    registerLayer("SeparableConv2D", SeparableConv2DCreator);
    registerLayer("Dense", DenseCreator);
    registerLayer("Subpixel", SubpixelCreator);
    registerLayer("Add", AddCreator);
    registerLayer("Activation", ActivationCreator);
    registerLayer("BatchNormalization", BatchNormalizationCreator);
    registerLayer("MaxPooling2D", MaxPooling2DCreator);
    registerLayer("AveragePooling2D", AveragePooling2DCreator);
    registerLayer("AdaptiveAvgPool2d", AdaptiveAvgPool2dCreator);
    registerLayer("Flatten", FlattenCreator);
    registerLayer("Pad", PadCreator);
    registerLayer("InstanceNorm", InstanceNormCreator);
    registerLayer("UpSampling2D", UpSampling2DCreator);
    registerLayer("Concatenate", ConcatenateCreator);
    registerLayer("Unary", UnaryCreator);
    registerLayer("Calculate", CalculateCreator);
    registerLayer("Conv2DTranspose", Conv2DTransposeCreator);
    registerLayer("YOLO", YOLOCreator);
}

GenericModelLayer* snn::dp::createLayerInstance(std::string layerName, ModelParser& parser, int i, bool useVulkan) { // layerFactory.cpp:136-159
    if (layerName == "DepthwiseConv2D" || layerName == "Depthwise") layerName = "SeparableConv2D";
    if (layerName == "subpixel" || layerName == "depth_to_space") layerName = "Subpixel";
    if (layerName == "InstanceNormalization") layerName = "InstanceNorm";
    if (layerName == "ZeroPadding2D") layerName = "Pad";
    LayerCreator creator = nullptr;
    {
        std::lock_guard<std::mutex> lock(registryMutex);
        auto it = LayerRegistryDict.find(layerName);
        if (it != LayerRegistryDict.end()) creator = it->second;
    }
    if (!creator) SNN_RIP("Not found layer: %s", layerName.c_str());
    return creator(parser, i, useVulkan);
}
