// hipBackend.cpp -- DeviceBackend / RenderPass on top of the C-ABI (counterpart of reference core/src/ic2/vulkanBackend.cpp
// and vulkanRenderpass.cpp).  No command buffer: plans enqueue kernels on the context's HIP stream, sync() waits for it.
#include <sys/stat.h>

#include "../../include/snnhip.h"
#include "ic2/backend.h"
#include "ic2/genericlayer.h"
#include "snn/contextFactory.h"

using namespace snn;
using namespace snn::dp;

static void hipChk(int rc, const char* what) {
    if (rc != SNNHIP_OK) SNN_RIP("%s: %s", what, snnhip_last_error());
}

HipRenderPass::~HipRenderPass() {
    if (owns && plan) snnhip_plan_destroy(plan);
}

void HipRenderPass::run() {
    if (skip) return;
    if (extraInputs.empty()) {
        hipChk(snnhip_plan_run(plan, input->tensor(), output->tensor()), "snnhip_plan_run");
    } else {
        std::vector<const snnhip_tensor*> ins{input->tensor()};
        for (auto* t : extraInputs) ins.push_back(t->tensor());
        hipChk(snnhip_plan_run_n(plan, ins.data(), static_cast<int>(ins.size()), output->tensor()), "snnhip_plan_run_n");
    }
}

bool HipRenderPass::debugPassOutput(const std::string& folder) { // vulkanBackend.cpp:132-134 naming
    if (skip) return true;
    output->saveToBIN(formatString("%s/%s pass[0].dump", folder.c_str(), name.c_str()));
    return true;
}

bool HipRenderPass::debugPassInputs(const std::string& folder) { // vulkanRenderpass.cpp:262-277 naming
    if (skip) return true;
    input->saveToBIN(formatString("%s/%s pass[0]_input.dump", folder.c_str(), name.c_str()));
    return true;
}

HipBackend::HipBackend(GpuContext* context) {
    SNN_CHK(context && context->backendType == GpuBackendType::HIP);
    ctx = static_cast<HipContext*>(context)->ctx;
}

HipBackend::~HipBackend() {
    dropRecording();
    for (auto* p : chainPlans) snnhip_plan_destroy(p);
    replacedPasses.clear();
}

// counterpart of VulkanBackend::initRenderPasses -> VulkanRenderPass ctor (vulkanBackend.cpp:43-78, vulkanRenderpass.cpp:103-178):
// this is where weights go to the device.
void HipBackend::initRenderPasses(GenericModelLayer* layer, ImageTextureArrayAccessor in, ImageTextureArrayAccessor out) {
    const InferencePasses* passes = layer->getPasses();
    SNN_CHK(passes && !passes->passes.empty());
    SNN_CHK(in.size() >= 1 && out.size() >= 1);
    auto& rps = layer->getRenderPasses();
    rps.clear();
    for (const auto& pass : passes->passes) {
        snnhip_plan* plan = nullptr;
        hipChk(pass.createPlan(ctx, &plan), pass.source.c_str());
        int od[4];
        hipChk(snnhip_plan_output_dims(plan, od), "snnhip_plan_output_dims");
        ImageTexture& o = out[0];
        // Dense produces [1][1][1][Out] while the reference's texture is Out x 1 x 1: accept both, reject real mismatches
        const size_t want = static_cast<size_t>(od[0]) * od[1] * od[2] * od[3];
        const size_t have = static_cast<size_t>(o.width()) * o.height() * o.channels();
        if (want != have) SNN_RIP("%s: plan output %dx%dx%dx%d does not match texture %s", layer->getName().c_str(), od[0], od[1], od[2], od[3], o.getTextureInfo2().c_str());
        auto rp = std::make_shared<HipRenderPass>(plan, &in[0], &o, layer->getName(), true);
        for (size_t k = 1; k < in.size(); ++k) rp->extraInputs.push_back(&in[k]);
        rps.push_back(rp);
        char buf[256];
        snnhip_plan_describe(plan, buf, sizeof(buf));
        SNN_LOGD("%s -> %s", layer->getName().c_str(), buf);
    }
}

bool HipBackend::sync() {
    hipChk(snnhip_sync(ctx), "snnhip_sync");
    return true;
}

bool HipBackend::beginRecord() {
    dropRecording();
    return snnhip_graph_begin_capture(ctx) == SNNHIP_OK;
}

bool HipBackend::endRecord() {
    snnhip_graph* g = nullptr;
    if (snnhip_graph_end_capture(ctx, &g) != SNNHIP_OK) return false;
    recording = g;
    return true;
}

bool HipBackend::replay() { return recording && snnhip_graph_launch(static_cast<snnhip_graph*>(recording)) == SNNHIP_OK; }

void HipBackend::dropRecording() {
    if (recording) snnhip_graph_destroy(static_cast<snnhip_graph*>(recording));
    recording = nullptr;
}

DeviceTimer* HipBackend::createDeviceTimer(const std::string& name) { return new HipDeviceTimer(ctx, name); }

void HipBackend::postRun(RenderStagesArray&, bool dumpOutput, const std::string& folder) {
    (void) folder;
    (void) dumpOutput; // dumps are written by the layers' render passes right after they run (GenericModelLayer::run)
}

// Replace linear runs of single-pass stages by fused plans (snnhip_chain_plan_create).  A stage can join a run when it has
// exactly one input, that input is the previous stage's output, and nobody else consumes that output.
void HipBackend::finalizeStages(RenderStagesArray& stages, bool dumpOutputs, bool fuseChains) {
    if (dumpOutputs || !fuseChains) return; // dumps need every intermediate tensor
    std::vector<int> consumers(stages.size(), 0);
    for (auto& s : stages)
        for (size_t j = 0; j < s.inputIds.size(); ++j)
            if (!s.delayBindMask[j] && s.inputIds[j] >= 0) consumers[static_cast<size_t>(s.inputIds[j])]++;
    auto passOf = [&](size_t i) -> HipRenderPass* {
        auto* ml = static_cast<GenericModelLayer*>(stages[i].layer->modelLayer);
        if (!ml || stages[i].layer->isInputLayer || ml->getRenderPasses().size() != 1) return nullptr;
        return dynamic_cast<HipRenderPass*>(ml->getRenderPasses()[0].get());
    };
    size_t i = 0;
    while (i < stages.size()) {
        if (!passOf(i) || stages[i].inputIds.size() != 1) { // a chain starts at a single-input stage (Add has two)
            ++i;
            continue;
        }
        size_t j = i;
        while (j + 1 < stages.size() && passOf(j + 1) && stages[j + 1].inputIds.size() == 1 && !stages[j + 1].delayBindMask[0] &&
               stages[j + 1].inputIds[0] == static_cast<int>(j) && consumers[j] == 1)
            ++j;
        if (j > i) {
            std::vector<snnhip_plan*> plans;
            for (size_t k = i; k <= j; ++k) plans.push_back(passOf(k)->plan);
            snnhip_plan* chain = nullptr;
            const int rc = snnhip_chain_plan_create(ctx, plans.data(), static_cast<int>(plans.size()), &chain);
            if (rc == SNNHIP_OK) {
                chainPlans.push_back(chain);
                HipRenderPass* first = passOf(i);
                HipRenderPass* last = passOf(j);
                // the first stage launches the whole chain straight into the last stage's output tensor
                auto* ml = static_cast<GenericModelLayer*>(stages[i].layer->modelLayer);
                replacedPasses.push_back(ml->getRenderPasses()[0]);
                ml->getRenderPasses()[0] = std::make_shared<HipRenderPass>(chain, first->input, last->output, first->name + " (+fused chain)", false);
                for (size_t k = i + 1; k <= j; ++k) {
                    passOf(k)->skip = true;
                    stages[k].fusedAway = true;
                }
                char buf[512];
                snnhip_plan_describe(chain, buf, sizeof(buf));
                SNN_LOGI("stages %zu..%zu fused: %s", i, j, buf);
            } else if (rc != SNNHIP_E_UNSUPPORTED) {
                hipChk(rc, "snnhip_chain_plan_create");
            }
        }
        i = j + 1;
    }
}

DeviceBackend* BackendBuilder::build(GpuContext* context, const InferenceGraph&) { // backendBuilder.cpp:36-58
    SNN_ASSERT(context);
    switch (context->backendType) {
    case GpuBackendType::HIP: return new HipBackend(context);
    default: SNN_CHK(false);
    }
    return nullptr;
}
