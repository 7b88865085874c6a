// snn/inferencegraph.h -- high-level processing graph (reference core/inc/snn/inferencegraph.h:31-100), GPU_HIP added.
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "snn/color.h"
#include "snn/imageTexture.h"

namespace snn {
namespace dp {
class DeviceBackend;
}
struct InferenceGraph {
    enum class LayerExecutionType { CPU = 0, GPU_FS, GPU_CS, GPU_VK, GPU_HIP, NOT_DEFINED = 200 };
    struct IODesc {
        ColorFormat format;
        uint32_t width, height, depth, channels;
        uint32_t batch = 1; // HIP extension: images per inference (the reference fixes the 4th texture dim to 1, core.cpp:371)
    };
    struct LayerRef {
        bool isStageOutput = false;
        int index = -1;
    };
    struct Transform {
        bool isFixed = false;
        union {
            struct {
                float scaleWidth, scaleHeight, translateWidth, translateHeight;
            };
            struct {
                uint32_t fixedWidth, fixedHeight, fixedDepth, fixedBatch;
            };
        };
        static Transform identity() {
            Transform t;
            t.isFixed = false;
            t.scaleWidth = t.scaleHeight = 1.0f;
            t.translateWidth = t.translateHeight = 0.0f;
            return t;
        }
    };
    struct Layer {
        LayerExecutionType layerLoc = LayerExecutionType::NOT_DEFINED;
        std::string name;
        std::vector<LayerRef> inputRefs;
        IODesc outputDesc{};
        bool flattenLayer = false;
        bool isInputLayer = false;
        uint32_t inputIndex = 0;
        using TImageTextureFunc = std::function<void(ImageTextureArray&, ImageTextureArray&)>;
        TImageTextureFunc imageTextureFunPtr;
        using TInitFunc = std::function<void(snn::dp::DeviceBackend*, ImageTextureArray&, ImageTextureArray&)>;
        TInitFunc initFunPtr;
        using TRunFunc = std::function<void(snn::dp::DeviceBackend*, bool)>;
        TRunFunc runFunPtr;
        void* modelLayer = nullptr; // GenericModelLayer* (host mirror only: lets the backend fuse neighbouring plans)
    };
    std::vector<IODesc> inputsDesc;
    std::vector<std::shared_ptr<Layer>> layers;
    MRTMode mrtMode = MRTMode::DOUBLE_PLANE;
    WeightAccessMethod weightMode = WeightAccessMethod::TEXTURES;
};
} // namespace snn
