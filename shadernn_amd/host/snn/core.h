// snn/core.h -- MixedInferenceCore / RenderStage (reference core/inc/snn/core.h:37-146).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "snn/deviceTimer.h"
#include "snn/inferencegraph.h"
#include "snn/layeroption.h"
#include "snn/snn.h"

namespace snn {
namespace dp {
class DeviceBackend;
}
typedef enum class Transition { Backend_CPU_GPU, Backend_GPU_CPU, NOT_DEFINED = 200 } Transition;

struct RenderStage {
    explicit RenderStage(GpuContext* context) : stageInputs(context), stageOutputs(context) {}
    std::shared_ptr<InferenceGraph::Layer> layer;
    bool flattenLayer = false;
    std::shared_ptr<DeviceTimer> timer;
    Backend backend = Backend::Backend_GPU;
    Transition transition = Transition::NOT_DEFINED;
    ImageTextureArray stageInputs;
    ImageTextureArray stageOutputs;
    std::vector<int> inputIds;
    std::vector<int> delayBindMask;
    bool fusedAway = false; // HIP extension: this stage's work is done by the fused plan of a later stage (HipBackend::finalizeStages)
    // HIP extension (HipBackend::finalizeStages): this stage and the previous launching stage read only tensors produced earlier -- two
    // branches of a residual block (ResNet: the 3x3 stride-2 convolution and the 1x1 stride-2 downsample of the same input).  run() issues this
    // one on the backend's side stream next to the previous one instead of behind it (DeviceBackend::forkSide / joinSide).
    bool sideOfPrevious = false;
    // ... or, when both stages run plans the library can launch as ONE grid (snnhip_plan_groupable: the K-split convolutions), inside a launch group
    // (DeviceBackend::groupBegin / groupEnd): the default for such pairs; SNN_STAGE_GROUPS=0 switches it off
    bool groupWithPrevious = false;
};
typedef std::vector<RenderStage> RenderStagesArray;

class MixedInferenceCore {
public:
    virtual ~MixedInferenceCore();
    SNN_NO_COPY(MixedInferenceCore);
    struct RunParameters {
        ImageTextureArray* inputImages = nullptr;
        ImageTextureArray* outputImages = nullptr;
        std::vector<std::vector<std::vector<float>>> inputMatrix;
        std::vector<std::vector<std::vector<float>>> output;
        SNNModelOutput modelOutput;
        // HIP extension: return right after the inference is enqueued instead of waiting for it (the reference waits once per inference,
        // core.cpp:203).  The caller keeps several inferences in flight on the backend's stream and calls sync() when it needs a result;
        // ignored with profiling / dumps / CPU stages / a classifier or detection output (those read results on the host inside run()).
        bool deferSync = false;
    };
    void run(RunParameters& rp);
    bool sync(); // HIP extension: wait for every inference enqueued with deferSync
    struct CreationParameters : InferenceGraph {
        uint32_t outputWidth = 0, outputHeight = 0, outputDepth = 0;
        bool dumpOutputs = false;
        bool fuseChains = true;
        bool profiling = false; // the reference enables per-stage timers at compile time (-DPROFILING, CMakeLists.txt:44-46)
        // HIP extension: record the launch sequence of the first run() as a hipGraph and replay it while the caller keeps feeding the same input
        // textures (re-recorded when they change).  Ignored with dumpOutputs / profiling / CPU stages (those need the host between launches).
        bool captureGraph = false;
    };
    static std::unique_ptr<MixedInferenceCore> create(GpuContext* context, const CreationParameters& cp);
    static std::unique_ptr<MixedInferenceCore> create(GpuContext* context, const std::string& modelFileName, const dp::ShaderGenOptions& options,
                                                      bool dumpOutputs = false);
    void writeTimeStat(std::map<std::string, std::vector<double>>& timeArray);
    size_t numStages() const { return stages.size(); }
    RenderStage& stage(size_t i) { return stages[i]; }
    std::string describe() const; // HIP extension: which kernel variant each stage runs
    // HIP extension: run the next inferences launch by launch even when a recording exists (per-launch profiling needs the plans to run)
    void suspendReplay(bool suspend) { replaySuspended = suspend; }

private:
    GpuContext* context;
    bool bindOutput = true;
    CreationParameters cp;
    RenderStagesArray stages;
    dp::DeviceBackend* backend = nullptr;
    DeviceTimer* gpuRunTime = nullptr;
    struct InputKey { // what a recorded launch sequence depends on: the device buffer (address), its extent and element type
        const void* data;
        int dims[4];
        int dtype;
        bool operator==(const InputKey& o) const {
            return data == o.data && dims[0] == o.dims[0] && dims[1] == o.dims[1] && dims[2] == o.dims[2] && dims[3] == o.dims[3] && dtype == o.dtype;
        }
    };
    std::vector<InputKey> recordedInputs; // device buffers the recording reads (delay-bound model inputs)
    bool graphUsable = false;
    bool replaySuspended = false;
    Timer cpuRunTime = Timer("IC2 Total CPU Runtime");
    explicit MixedInferenceCore(GpuContext* context_);
    bool init(const CreationParameters& cp);
};
} // namespace snn
