// ic2/dp.h -- "dynamic pipeline": JSON -> layer DAG -> InferenceGraph (reference core/src/ic2/dp.{h,cpp}).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "ic2/genericlayer.h"

namespace snn {
namespace dp {
typedef std::vector<std::shared_ptr<GenericModelLayer>> InferenceModel;
InferenceModel loadFromJsonModel(const std::string& fileName, bool useVulkan, const MRTMode& mrtMode, const WeightAccessMethod& weightMode,
                                 bool preferHp = false);
// multi-input form (dp.cpp:432-640); the single-head form forwards to it
InferenceGraph generateInferenceGraph(std::vector<std::shared_ptr<GenericModelLayer>>& layers, const ShaderGenOptions& options);
InferenceGraph generateInferenceGraph(const std::shared_ptr<GenericModelLayer> firstLayer, const ShaderGenOptions& options);
} // namespace dp
} // namespace snn
