// snn/snn.h -- main enumerations and base classes (reference core/inc/snn/snn.h:37-115), plus the HIP backend type.
#pragma once
#include <vector>

#include "snn/utils.h"

#ifndef OUTPUT_DIR
#define OUTPUT_DIR "inferenceCoreDump" // reference: "../../../../core/inferenceCoreDump" (snn.h:39); overridable via env SNN_OUTPUT_DIR
#endif

#define DIV_4_ROUND_UP(i) (((i) + 3) / 4)
#define UP_DIV(x, y) (((x) + (y) - (1)) / (y))
#define ROUND_UP(x, y) (((x) + (y) - (1)) / (y) * (y))

namespace snn {

const char* outputDir(); // OUTPUT_DIR or $SNN_OUTPUT_DIR

enum Device { GPU, CPU };
enum class GpuBackendType { GL, VULKAN, HIP }; // GL / VULKAN kept for source compatibility; only HIP is implemented here
enum class Precision { FP32, FP16 };
enum class WeightAccessMethod { CONSTANTS, TEXTURES, UNIFORM_BUFFER, SSBO_BUFFER };
enum class MRTMode { NO = 0, SINGLE_PLANE = 4, DOUBLE_PLANE = 8, QUAD_PLANE = 16 };

class GpuContext {
public:
    const GpuBackendType backendType;
    SNN_NO_COPY(GpuContext);
    SNN_NO_MOVE(GpuContext);
    virtual ~GpuContext() = default;

protected:
    explicit GpuContext(GpuBackendType t) : backendType(t) {}
};

enum ModelType { CLASSIFICATION, DETECTION, SEGMENTATION, OTHER };

struct SNNModelOutput {
    ModelType modelType = ModelType::OTHER;
    int classifierOutput = 0;
    std::vector<std::vector<float>> detectionOutput;
};

} // namespace snn
