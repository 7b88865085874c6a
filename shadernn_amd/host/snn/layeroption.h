// snn/layeroption.h -- reference core/inc/snn/layeroption.h:27-48 (fields kept; `vulkan`/`compute` are accepted and ignored:
// every GPU layer runs through the HIP plans).
#pragma once
#include <vector>

#include "snn/inferencegraph.h"
#include "snn/snn.h"
namespace snn {
namespace dp {
struct ShaderGenOptions {
    std::vector<InferenceGraph::IODesc> desiredInput;
    ColorFormat desiredOutputFormat = ColorFormat::RGBA32F;
    bool compute = false;
    bool vulkan = false;
    bool preferrHalfPrecision = false;
    bool ssbo = false;
    MRTMode mrtMode = MRTMode::SINGLE_PLANE;
    WeightAccessMethod weightMode = WeightAccessMethod::TEXTURES;
    // HIP backend extension: fuse linear runs of plans (snnhip_chain_plan_create) when outputs are not dumped
    bool fuseChains = true;
    // HIP backend extension: images per inference; every stage tensor is [batch][H][W][C] (the reference runs one image, core.cpp:371)
    uint32_t batch = 1;
};
} // namespace dp
} // namespace snn
