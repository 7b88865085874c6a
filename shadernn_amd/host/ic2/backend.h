// ic2/backend.h -- DeviceBackend / RenderPass / InferencePass (reference core/src/ic2/backend.h:32-91, renderpass.h:24-67,
// inferencepass.h:31-62) and the HIP implementations that sit on the C-ABI of include/snnhip.h.
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "snn/core.h"
#include "snn/deviceTimer.h"
#include "snn/imageTexture.h"

struct snnhip_plan;
struct snnhip_ctx;

namespace snn {

class RenderPass {
public:
    RenderPass() = default;
    virtual ~RenderPass() = default;
    SNN_NO_COPY(RenderPass);
    virtual bool debugPassOutput(const std::string&) { return true; }
    virtual bool debugPassInputs(const std::string&) { return true; }
    virtual bool debugPassWeights(const std::string&, int) { return true; }
    virtual void run() {}
};

// One inference pass = what the layer's createCS() produced.  The Vulkan flavour carries SPIR-V + specialisation constants
// (inferencepassVulkan.h:27-41); the HIP flavour carries an executable plan (device-resident weights + launch recipe).
struct InferencePass {
    std::string source; // operator description (the reference stores the shader asset name here)
    // Host-side recipe: packed descriptor + weights captured by value.  The backend turns it into device resources in
    // initRenderPasses(), exactly where VulkanRenderPass's constructor builds the pipeline and uploads the weight image.
    std::function<int(snnhip_ctx*, snnhip_plan**)> createPlan;
};
struct InferencePasses {
    std::vector<InferencePass> passes;
};
typedef std::shared_ptr<InferencePasses> InferencePassesSptr;

namespace dp {

class GenericModelLayer;

class DeviceBackend {
public:
    DeviceBackend() = default;
    virtual ~DeviceBackend() = default;
    SNN_NO_COPY(DeviceBackend);
    virtual void initRenderPasses(GenericModelLayer*, ImageTextureArrayAccessor, ImageTextureArrayAccessor) {}
    virtual void prepareRun(MixedInferenceCore::RunParameters&, RenderStagesArray&, bool, uint32_t) {}
    virtual void prepareStage(MixedInferenceCore::RunParameters&, RenderStage&) {}
    virtual void postRun(RenderStagesArray&, bool, const std::string&) {}
    virtual bool sync() { return false; }
    virtual void cleanupRun() {}
    virtual DeviceTimer* createDeviceTimer(const std::string&) { return nullptr; }
    virtual bool isProfilingEnabled(bool = false) { return true; }
    // HIP extension, called once after every stage is initialised: may replace linear runs of passes by fused plans
    virtual void finalizeStages(RenderStagesArray&, bool /*dumpOutputs*/, bool /*fuseChains*/) {}
    // HIP extension: record the launches of one inference once and replay them (the counterpart of the Vulkan backend recording a command
    // buffer, vulkanBackend.cpp:80-95).  beginRecord returns false when the backend cannot record; replay re-submits the last recording.
    virtual bool beginRecord() { return false; }
    virtual bool endRecord() { return false; }
    virtual bool replay() { return false; }
    virtual void dropRecording() {}
    // HIP extension: two independent stages side by side.  forkSide: what runs next goes to a second stream that starts behind the work enqueued
    // so far; backToMain: what runs next goes to the main stream again; joinSide: the main stream waits for the side stream.
    virtual bool forkSide() { return false; }
    virtual void backToMain() {}
    virtual void joinSide() {}
    // HIP extension: a launch group -- groupable plans run between the two calls are launched by groupEnd, two compatible ones as one kernel launch
    virtual bool groupBegin() { return false; }
    virtual void groupEnd() {}
};

class HipRenderPass : public RenderPass {
public:
    HipRenderPass(snnhip_plan* p, ImageTexture* in, ImageTexture* out, const std::string& layerName, bool ownsPlan = false)
        : plan(p), input(in), output(out), name(layerName), owns(ownsPlan) {}
    std::vector<ImageTexture*> extraInputs; // inputs 1.. of multi-input operators (Add: addlayerVulkan.cpp:98 binds uInput0, uInput1)
    ~HipRenderPass() override;
    void run() override;                                   // enqueue only (vulkanRenderpass.cpp:257-259 records Dispatch + barrier)
    bool debugPassOutput(const std::string& folder) override; // "<folder>/<layer name> pass[0].dump" (vulkanBackend.cpp:132-134)
    bool debugPassInputs(const std::string& folder) override; // "<layer name> pass[0]_input.dump" (vulkanRenderpass.cpp:262-277)
    snnhip_plan* plan;
    ImageTexture *input, *output;
    std::string name;
    bool owns;
    bool skip = false; // fused into an earlier pass
};

class HipBackend : public DeviceBackend {
public:
    explicit HipBackend(GpuContext* context);
    void initRenderPasses(GenericModelLayer* layer, ImageTextureArrayAccessor in, ImageTextureArrayAccessor out) override;
    void postRun(RenderStagesArray& stages, bool dumpOutput, const std::string& folder) override;
    bool sync() override; // hipStreamSynchronize == QueueSubmitAndWait (vulkanBackend.cpp:97-106)
    DeviceTimer* createDeviceTimer(const std::string& name) override;
    void finalizeStages(RenderStagesArray& stages, bool dumpOutputs, bool fuseChains) override;
    bool beginRecord() override;
    bool endRecord() override;
    bool replay() override;
    void dropRecording() override;
    bool forkSide() override;
    void backToMain() override;
    void joinSide() override;
    bool groupBegin() override;
    void groupEnd() override;

private:
    snnhip_ctx* ctx;
    std::vector<snnhip_plan*> chainPlans; // owned
    void* recording = nullptr;            // snnhip_graph* of the last recorded inference
    std::vector<std::shared_ptr<RenderPass>> replacedPasses; // their plans may still be referenced by a chain (unfused steps)
public:
    ~HipBackend() override;
};

struct BackendBuilder {
    static DeviceBackend* build(GpuContext* context, const InferenceGraph& ig); // backendBuilder.cpp:36-58
};

} // namespace dp
} // namespace snn
