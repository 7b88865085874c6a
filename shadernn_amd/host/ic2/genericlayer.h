// ic2/genericlayer.h -- layer base classes and the per-operator layer definitions of the hot path
// (reference core/src/ic2/genericlayer.h:36-197, conv2d.h, separableconvolution.h, denselayer.h, subpixelmerge.h, inputlayer.h).
// createCS() builds an executable HIP plan through the C-ABI instead of a SPIR-V pass.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ic2/backend.h"
#include "ic2/modelparser.h"
#include "snn/inferencegraph.h"
#include "snn/layeroption.h"

namespace snn {
namespace dp {

struct CommonLayerDesc {
    bool isRange01 = false;
    uint32_t numOutputPlanes = 0;
    uint32_t numInputPlanes = 0;
    uint32_t kernelSize = 0;
    MRTMode mrtMode = MRTMode::NO;
    WeightAccessMethod weightMode = WeightAccessMethod::CONSTANTS;
    bool preferHp = false;
    bool isInputLayer = false;
    uint32_t inputIndex = 0;
    void parse(ModelParser& parser, int layerId) {
        isRange01 = parser.isInputRange01();
        numOutputPlanes = static_cast<uint32_t>(parser.getOutputPlanes(layerId));
        numInputPlanes = static_cast<uint32_t>(parser.getInputPlanes(layerId));
        preferHp = parser.getPrecision();
        mrtMode = parser.getMRTMode();
        weightMode = parser.getWeightMode();
    }
};

class GenericModelLayer {
public:
    struct LayerGenOptions : ShaderGenOptions {
        uint32_t desiredOutputWidth = 0, desiredOutputHeight = 0;
        bool isFirstLayer = false, isLastLayer = false;
    };
    explicit GenericModelLayer(CommonLayerDesc d) : _desc(d) {}
    GenericModelLayer(const GenericModelLayer&) = delete;
    GenericModelLayer& operator=(const GenericModelLayer&) = delete;
    virtual ~GenericModelLayer();

    const CommonLayerDesc& getDesc() const { return _desc; }
    const std::string& getName() const { return name; }
    void setName(const std::string& n) { name = n; }
    const InferencePasses* getPasses() const { return passes.get(); }
    std::vector<std::shared_ptr<RenderPass>>& getRenderPasses() { return renderPasses; }
    bool isInputLayer() const { return _desc.isInputLayer; }
    uint32_t getInputIndex() const { return _desc.inputIndex; }
    void addInputDim(const InferenceGraph::IODesc& dim) { inputDims.push_back(dim); }
    void setMRTMode(const MRTMode& m) { _desc.mrtMode = m; }
    void setWeightAccessMode(const WeightAccessMethod& m) { _desc.weightMode = m; }

    virtual void init(DeviceBackend* backend, ImageTextureArray& inputMat, ImageTextureArray& outputMat);
    virtual void run(DeviceBackend* backend, bool dumpOutputs);
    virtual void getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const;
    virtual void computeImageTexture(ImageTextureArray&, ImageTextureArray&) {}
    virtual void createInferencePasses(const LayerGenOptions& options) = 0;
    virtual InferenceGraph::LayerExecutionType getLayerExecutionType() const = 0;
    virtual void setLayerExecutionType(InferenceGraph::LayerExecutionType) = 0;

    std::vector<std::shared_ptr<GenericModelLayer>> prevLayers;
    std::vector<std::shared_ptr<GenericModelLayer>> nextLayers;

protected:
    std::string name;
    CommonLayerDesc _desc;
    std::vector<InferenceGraph::IODesc> inputDims;
    InferencePassesSptr passes;
    std::vector<std::shared_ptr<RenderPass>> renderPasses;

private:
    virtual InferenceGraph::Transform getOutputScaleDimAdjustment() const = 0;
};

struct GenericConvDesc : CommonLayerDesc {
    std::vector<WeightMat> weightsCvM; // OC*IC (depthwise: C) matrices of k x k; the field name the reference tests fill
    std::vector<double> biases;
    std::string activation;
    uint32_t stride = 0;
    void parse(ModelParser& parser, int layerId) { CommonLayerDesc::parse(parser, layerId); }
};

class ShaderLayer : public GenericModelLayer {
public:
    explicit ShaderLayer(CommonLayerDesc d) : GenericModelLayer(d) {}
    void createInferencePasses(const LayerGenOptions& options) override; // genericlayer.cpp:92-114
    InferenceGraph::LayerExecutionType getLayerExecutionType() const override { return executeBackend; }
    void setLayerExecutionType(InferenceGraph::LayerExecutionType e) override { executeBackend = e; }

private:
    InferenceGraph::Transform getOutputScaleDimAdjustment() const override { return InferenceGraph::Transform::identity(); }
    virtual InferencePassesSptr createCS(const LayerGenOptions&) const = 0;
    InferenceGraph::LayerExecutionType executeBackend = InferenceGraph::LayerExecutionType::GPU_HIP;
};

// ---- InputLayer (inputlayer.h)
struct InputLayerDesc : CommonLayerDesc {
    uint32_t inputHeight = 0, inputWidth = 0, inputChannels = 0;
    void parse(ModelParser& parser, int layerId) {
        CommonLayerDesc::parse(parser, layerId);
        parser.getInputLayer(layerId, inputWidth, inputHeight, inputChannels, inputIndex);
        isInputLayer = true;
    }
};
class InputLayerLayer : public ShaderLayer {
public:
    explicit InputLayerLayer(InputLayerDesc d) : ShaderLayer(d), desc(d) {}

private:
    InferencePassesSptr createCS(const LayerGenOptions&) const override { SNN_RIP("Not implemented !"); }
    InputLayerDesc desc;
};

// ---- Conv2D (conv2d.h, conv2d.cpp, conv2dVulkan.cpp)
struct Conv2DDesc : GenericConvDesc {
    bool useBatchNormalization = false;
    bool useMultiInputs = false;
    std::map<std::string, std::vector<float>> batchNormalization;
    float leakyReluAlpha = 0.0f;
    std::string padding;
    bool useUniformShaders = true;
    std::string paddingT, paddingB, paddingL, paddingR;
    std::string paddingMode = "constant";
    void parse(ModelParser& parser, int layerId);
};
class Conv2DLayer : public ShaderLayer {
public:
    explicit Conv2DLayer(Conv2DDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}
    void getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const override;
    const Conv2DDesc& convDesc() const { return _desc; }

protected:
    Conv2DDesc _desc;
    void getPaddingOffset(uint32_t (&offsets)[4]) const; // conv2d.cpp:39-74

private:
    InferenceGraph::Transform getOutputScaleDimAdjustment() const override; // conv2d.cpp:102-113
};
class Conv2DLayerHip : public Conv2DLayer {
public:
    explicit Conv2DLayerHip(Conv2DDesc&& d) : Conv2DLayer(std::move(d)) {}

private:
    InferencePassesSptr createCS(const LayerGenOptions&) const override;
};

// ---- SeparableConv2D == depthwise (separableconvolution.h/.cpp, separableconvolutionVulkan.cpp)
struct SeparableConv2DDesc : GenericConvDesc {
    bool useBatchNormalization = false;
    std::map<std::string, std::vector<float>> batchNormalization;
    float leakyReluAlpha = 0.0f;
    std::string paddingT, paddingB, paddingL, paddingR;
    void parse(ModelParser& parser, int layerId);
};
class SeparableConv2DLayer : public ShaderLayer {
public:
    explicit SeparableConv2DLayer(SeparableConv2DDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}
    void getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const override; // separableconvolution.cpp:77-86

protected:
    SeparableConv2DDesc _desc;
    void getPaddingOffset(uint32_t (&offsets)[4]) const;

private:
    InferenceGraph::Transform getOutputScaleDimAdjustment() const override;
};
class SeparableConv2DLayerHip : public SeparableConv2DLayer {
public:
    explicit SeparableConv2DLayerHip(SeparableConv2DDesc&& d) : SeparableConv2DLayer(std::move(d)) {}

private:
    InferencePassesSptr createCS(const LayerGenOptions&) const override;
};

// ---- Dense (denselayer.h/.cpp, denselayerVulkan.cpp, cpulayer.h)
struct DenseDesc : CommonLayerDesc {
    std::vector<std::vector<float>> weights; // [In][Out]-shaped rows of the FLAT kernel; consumed flat as [Out][In]
    std::vector<float> biases;
    std::string activation;
    int numInputUnits = 0, numOutputUnits = 0;
    float leakyReluAlpha = 0.0f;
    void parse(ModelParser& parser, int layerId) {
        CommonLayerDesc::parse(parser, layerId);
        parser.getDenseLayer(layerId, numOutputUnits, numInputUnits, activation, weights, biases, leakyReluAlpha);
    }
};
class DenseLayer : public ShaderLayer {
public:
    explicit DenseLayer(DenseDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}
    void getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const override; // denselayer.cpp:51-55

protected:
    DenseDesc _desc;

private:
    InferenceGraph::Transform getOutputScaleDimAdjustment() const override;
};
class DenseLayerHip : public DenseLayer {
public:
    explicit DenseLayerHip(DenseDesc&& d) : DenseLayer(std::move(d)) {}

private:
    InferencePassesSptr createCS(const LayerGenOptions&) const override;
};

// ---- Subpixel (subpixelmerge.h, subpixelmergeVulkan.cpp)
struct SubpixelDesc : CommonLayerDesc {
    uint32_t kernelSize = 2;
    std::vector<double> biases;
    void parse(ModelParser& parser, int layerId) {
        CommonLayerDesc::parse(parser, layerId);
        kernelSize = 2; // hard-coded in the reference (subpixelmerge.h:26-33)
    }
};
class SubpixelLayer : public ShaderLayer {
public:
    explicit SubpixelLayer(SubpixelDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}

protected:
    SubpixelDesc _desc;

private:
    InferenceGraph::Transform getOutputScaleDimAdjustment() const override {
        InferenceGraph::Transform t = InferenceGraph::Transform::identity();
        t.scaleWidth = t.scaleHeight = static_cast<float>(_desc.kernelSize);
        return t;
    }
};
class SubpixelLayerHip : public SubpixelLayer {
public:
    explicit SubpixelLayerHip(SubpixelDesc&& d) : SubpixelLayer(std::move(d)) {}

private:
    InferencePassesSptr createCS(const LayerGenOptions&) const override;
};

// ------------------------------------------------------------------------------------------------------------------
// Operators between the convolutions.  Desc = the reference's struct (same fields / parse calls), one
// Layer base per operator with the reference's output-size rule, and a Hip flavour whose createCS() builds the plan.

// ---- Add (addlayer.h:27-46, addlayerVulkan.cpp:33-114)
struct AddDesc : CommonLayerDesc {
    std::string activation;
    float leakyReluAlpha = 0.0f;
    void parse(ModelParser& parser, int layerId) {
        CommonLayerDesc::parse(parser, layerId);
        parser.getAddLayer(layerId, activation, leakyReluAlpha);
    }
};
class AddLayer : public ShaderLayer {
public:
    explicit AddLayer(AddDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}

protected:
    AddDesc _desc;
};
class AddLayerHip : public AddLayer {
public:
    explicit AddLayerHip(AddDesc&& d) : AddLayer(std::move(d)) {}

private:
    InferencePassesSptr createCS(const LayerGenOptions&) const override;
};

// ---- Activation (activation.h, activationVulkan.cpp)
struct ActivationDesc : CommonLayerDesc {
    std::string activation;
    float leakyReluAlpha = 0.0f;
    void parse(ModelParser& parser, int layerId) {
        CommonLayerDesc::parse(parser, layerId);
        parser.getActivationLayer(layerId, activation, leakyReluAlpha);
    }
};
class ActivationLayer : public ShaderLayer {
public:
    explicit ActivationLayer(ActivationDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}

protected:
    ActivationDesc _desc;
};
class ActivationLayerHip : public ActivationLayer {
public:
    explicit ActivationLayerHip(ActivationDesc&& d) : ActivationLayer(std::move(d)) {}

private:
    InferencePassesSptr createCS(const LayerGenOptions&) const override;
};

// ---- BatchNormalization (batchnorm.h:27-50, batchnormVulkan.cpp)
struct BatchNormalizationDesc : GenericConvDesc {
    float leakyReluAlpha = 0.0f;
    std::map<std::string, std::vector<float>> batchNormalization;
    void parse(ModelParser& parser, int layerId) {
        GenericConvDesc::parse(parser, layerId);
        int oc = 0, ic = 0;
        parser.getBatchNormLayer(layerId, oc, ic, batchNormalization, activation, leakyReluAlpha);
        numOutputPlanes = static_cast<uint32_t>(oc);
        numInputPlanes = static_cast<uint32_t>(ic);
    }
};
class BatchNormalizationLayer : public ShaderLayer {
public:
    explicit BatchNormalizationLayer(BatchNormalizationDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}

protected:
    BatchNormalizationDesc _desc;
};
class BatchNormalizationLayerHip : public BatchNormalizationLayer {
public:
    explicit BatchNormalizationLayerHip(BatchNormalizationDesc&& d) : BatchNormalizationLayer(std::move(d)) {}

private:
    InferencePassesSptr createCS(const LayerGenOptions&) const override;
};

// ---- MaxPooling2D / AveragePooling2D / AdaptiveAvgPool2d (maxpool2d.h/.cpp, avgpool2d.h/.cpp, adaptiveavgpool2d.h)
struct MaxPooling2DDesc : GenericConvDesc {
    std::string padding, paddingValue, paddingT, paddingB, paddingL, paddingR;
    void parse(ModelParser& parser, int layerId) {
        GenericConvDesc::parse(parser, layerId);
        int oc = 0, ic = 0, k = 0, s = 0;
        parser.getMaxPoolLayer(layerId, oc, ic, k, s, padding, paddingValue, paddingT, paddingB, paddingL, paddingR);
        numOutputPlanes = static_cast<uint32_t>(oc);
        numInputPlanes = static_cast<uint32_t>(ic);
        kernelSize = static_cast<uint32_t>(k);
        stride = static_cast<uint32_t>(s);
    }
};
struct AveragePooling2DDesc : GenericConvDesc {
    std::string padding;
    void parse(ModelParser& parser, int layerId) {
        GenericConvDesc::parse(parser, layerId);
        int oc = 0, ic = 0, k = 0, s = 0;
        parser.getAvgPoolLayer(layerId, oc, ic, k, s, padding);
        numOutputPlanes = static_cast<uint32_t>(oc);
        numInputPlanes = static_cast<uint32_t>(ic);
        kernelSize = static_cast<uint32_t>(k);
        stride = static_cast<uint32_t>(s);
    }
};
struct AdaptiveAvgPool2dDesc : GenericConvDesc {
    std::string padding;
    int targetSize = 1;
    void parse(ModelParser& parser, int layerId) {
        GenericConvDesc::parse(parser, layerId);
        int oc = 0, ic = 0;
        parser.getAdaptiveAvgPoolLayer(layerId, oc, ic, targetSize);
        numOutputPlanes = static_cast<uint32_t>(oc);
        numInputPlanes = static_cast<uint32_t>(ic);
    }
};
class MaxPooling2DLayer : public ShaderLayer {
public:
    explicit MaxPooling2DLayer(MaxPooling2DDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}

protected:
    MaxPooling2DDesc _desc;

private:
    InferenceGraph::Transform getOutputScaleDimAdjustment() const override; // maxpool2d.cpp:26-36
};
class AveragePooling2DLayer : public ShaderLayer {
public:
    explicit AveragePooling2DLayer(AveragePooling2DDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}

protected:
    AveragePooling2DDesc _desc;

private:
    InferenceGraph::Transform getOutputScaleDimAdjustment() const override; // avgpool2d.cpp:20-29
};
class AdaptiveAvgPool2dLayer : public ShaderLayer {
public:
    explicit AdaptiveAvgPool2dLayer(AdaptiveAvgPool2dDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}
    void getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const override;

protected:
    AdaptiveAvgPool2dDesc _desc;
};
#define SNN_DECLARE_HIP_FLAVOUR(layer)                                                    \
    class layer##LayerHip : public layer##Layer {                                         \
    public:                                                                               \
        explicit layer##LayerHip(layer##Desc&& d) : layer##Layer(std::move(d)) {}         \
                                                                                          \
    private:                                                                              \
        InferencePassesSptr createCS(const LayerGenOptions&) const override;              \
    };
SNN_DECLARE_HIP_FLAVOUR(MaxPooling2D)
SNN_DECLARE_HIP_FLAVOUR(AveragePooling2D)
SNN_DECLARE_HIP_FLAVOUR(AdaptiveAvgPool2d)

// ---- Flatten (flattenlayer.h/.cpp: HWC order, optional activation; a CPU layer in the reference)
struct FlattenDesc : CommonLayerDesc {
    std::string activation;
    float leakyReluAlpha = 0.0f;
    void parse(ModelParser& parser, int layerId) {
        CommonLayerDesc::parse(parser, layerId);
        int oc = 0, ic = 0;
        parser.getFlattenLayer(layerId, oc, ic, activation);
        numOutputPlanes = static_cast<uint32_t>(oc);
        numInputPlanes = static_cast<uint32_t>(ic);
    }
};
class FlattenLayer : public ShaderLayer {
public:
    explicit FlattenLayer(FlattenDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}
    void getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const override; // flattenlayer.cpp:58-62

protected:
    FlattenDesc _desc;
};
SNN_DECLARE_HIP_FLAVOUR(Flatten)

// ---- Pad (padlayer.h/.cpp, padlayerVulkan.cpp)
struct PadDesc : GenericConvDesc {
    float constant = 0.0f;
    std::string paddingT, paddingB, paddingL, paddingR;
    std::string mode = "constant";
    void parse(ModelParser& parser, int layerId) {
        GenericConvDesc::parse(parser, layerId);
        int oc = 0, ic = 0;
        parser.getPaddingLayer(layerId, oc, ic, paddingT, paddingB, paddingL, paddingR, mode, constant);
        numOutputPlanes = static_cast<uint32_t>(oc);
        numInputPlanes = static_cast<uint32_t>(ic);
    }
};
class PadLayer : public ShaderLayer {
public:
    explicit PadLayer(PadDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}

protected:
    PadDesc _desc;
    void getPaddingOffset(uint32_t (&offsets)[4]) const; // padlayer.cpp:27-56

private:
    InferenceGraph::Transform getOutputScaleDimAdjustment() const override; // padlayer.cpp:58-67
};
SNN_DECLARE_HIP_FLAVOUR(Pad)

// ---- InstanceNorm (instancenorm.h, instancenormVulkan.cpp)
struct InstanceNormDesc : GenericConvDesc {
    std::map<std::string, std::vector<float>> instanceNormalization;
    float leakyReluAlpha = 0.0f;
    float epsilon = 1e-5f;
    void parse(ModelParser& parser, int layerId) {
        GenericConvDesc::parse(parser, layerId);
        int oc = 0, ic = 0;
        parser.getInstanceNormalizationLayer(layerId, oc, ic, epsilon, instanceNormalization, activation, leakyReluAlpha);
        numOutputPlanes = static_cast<uint32_t>(oc);
        numInputPlanes = static_cast<uint32_t>(ic);
    }
};
class InstanceNormLayer : public ShaderLayer {
public:
    explicit InstanceNormLayer(InstanceNormDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}

protected:
    InstanceNormDesc _desc;
};
SNN_DECLARE_HIP_FLAVOUR(InstanceNorm)

// ---- UpSampling2D (upsampling2d.h, upsampling2dVulkan.cpp)
struct UpSampling2DDesc : CommonLayerDesc {
    float scale = 2.0f;
    std::string interpolationType;
    void parse(ModelParser& parser, int layerId) {
        CommonLayerDesc::parse(parser, layerId);
        scale = parser.getUpSamplingScale(layerId);
        interpolationType = parser.getUpSampling2DInterpolation(layerId);
    }
};
class UpSampling2DLayer : public ShaderLayer {
public:
    explicit UpSampling2DLayer(UpSampling2DDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}

protected:
    UpSampling2DDesc _desc;

private:
    InferenceGraph::Transform getOutputScaleDimAdjustment() const override { // upsampling2d.h:41-44
        InferenceGraph::Transform t = InferenceGraph::Transform::identity();
        t.scaleWidth = t.scaleHeight = _desc.scale;
        return t;
    }
};
SNN_DECLARE_HIP_FLAVOUR(UpSampling2D)

// ---- SURVEY 8f rank 4 ----------------------------------------------------------------------------------------------------------

// ---- Concatenate (concatenation.h:27-43, concatenationVulkan.cpp:31-88)
struct ConcatenateDesc : CommonLayerDesc {};
class ConcatenateLayer : public ShaderLayer {
public:
    explicit ConcatenateLayer(ConcatenateDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}
    void getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depthOut) const override { // concatenation.h:33-37: depth = texel planes
        width = inputDims[0].width;
        height = inputDims[0].height;
        depthOut = inputDims[0].depth + inputDims[1].depth;
    }

protected:
    ConcatenateDesc _desc;
};
SNN_DECLARE_HIP_FLAVOUR(Concatenate)

// ---- Unary (unary.h:26-42, unaryVulkan.cpp:30-83): opType / opValue are never read from the model file
struct UnaryDesc : CommonLayerDesc {
    int32_t opType = 0;
    float opValue = 1.0f;
    void parse(ModelParser& parser, int layerId) { CommonLayerDesc::parse(parser, layerId); }
};
class UnaryLayer : public ShaderLayer {
public:
    explicit UnaryLayer(UnaryDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}

protected:
    UnaryDesc _desc;
};
SNN_DECLARE_HIP_FLAVOUR(Unary)

// ---- Calculate (calculation.h:23, calculationGL.cpp:28-57)
struct CalculateDesc : CommonLayerDesc {};
class CalculateLayer : public ShaderLayer {
public:
    explicit CalculateLayer(CalculateDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}

protected:
    CalculateDesc _desc;
};
SNN_DECLARE_HIP_FLAVOUR(Calculate)

// ---- Conv2DTranspose (deconv2d.h:21, deconv2dGL.h:31-66, deconv2dGL.cpp:282-355)
struct Conv2DTransposeDesc : Conv2DDesc {};
class Conv2DTransposeLayer : public ShaderLayer {
public:
    explicit Conv2DTransposeLayer(Conv2DTransposeDesc&& d) : ShaderLayer(d), _desc(std::move(d)) {}

protected:
    Conv2DTransposeDesc _desc;

private:
    InferenceGraph::Transform getOutputScaleDimAdjustment() const override { // deconv2dGL.cpp:345-355
        InferenceGraph::Transform t = InferenceGraph::Transform::identity();
        t.scaleWidth = t.scaleHeight = static_cast<float>(_desc.stride);
        t.translateWidth = t.translateHeight = _desc.paddingT == "same" ? 0.0f : static_cast<float>(_desc.kernelSize - _desc.stride);
        return t;
    }
};
SNN_DECLARE_HIP_FLAVOUR(Conv2DTranspose)

// ---- YOLO (yololayer.h:33-65, yololayer.cpp:27-226): the final layer of YOLOv3-tiny, decoded and NMS-ed on the CPU in the reference too
struct YOLODesc : CommonLayerDesc {
    void parse(ModelParser& parser, int layerId) { CommonLayerDesc::parse(parser, layerId); } // yololayer.cpp:165-175: input / output planes only
};
class YOLOLayer : public GenericModelLayer {
public:
    explicit YOLOLayer(YOLODesc&& d) : GenericModelLayer(d), _yoloDesc(std::move(d)) {}
    void getOutputDims(uint32_t& width, uint32_t& height, uint32_t& depth) const override { // yololayer.h:44-48
        width = 100 * 6; // max 100 bounding boxes
        height = 1;
        depth = 1;
    }
    void computeImageTexture(ImageTextureArray& inputMat, ImageTextureArray& outputMat) override;
    InferenceGraph::LayerExecutionType getLayerExecutionType() const override { return executeBackend; }
    void setLayerExecutionType(InferenceGraph::LayerExecutionType e) override { executeBackend = e; }
    void createInferencePasses(const LayerGenOptions&) override {}
    // decode + NMS on host floats, heads in NHWC with the true channel count (3 x 6); returns rows {class, score, x, y, w, h}
    static std::vector<std::vector<float>> decode(const std::vector<const float*>& heads, int netSize = 416);

private:
    InferenceGraph::Transform getOutputScaleDimAdjustment() const override { return InferenceGraph::Transform::identity(); }
    YOLODesc _yoloDesc;
    InferenceGraph::LayerExecutionType executeBackend = InferenceGraph::LayerExecutionType::CPU;
};

} // namespace dp
} // namespace snn
