// hipBackend.cpp -- DeviceBackend / RenderPass on top of the C-ABI (counterpart of reference core/src/ic2/vulkanBackend.cpp
// and vulkanRenderpass.cpp).  No command buffer: plans enqueue kernels on the context's HIP stream, sync() waits for it.
#include <sys/stat.h>

#include <cstdlib>

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
        const size_t have = static_cast<size_t>(o.batch()) * o.width() * o.height() * o.channels();
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

bool HipBackend::forkSide() { return snnhip_ctx_fork(ctx) == SNNHIP_OK; }
void HipBackend::backToMain() { hipChk(snnhip_ctx_main(ctx), "snnhip_ctx_main"); }
void HipBackend::joinSide() { hipChk(snnhip_ctx_join(ctx), "snnhip_ctx_join"); }
bool HipBackend::groupBegin() { return snnhip_ctx_group_begin(ctx) == SNNHIP_OK; }
void HipBackend::groupEnd() { hipChk(snnhip_ctx_group_end(ctx), "snnhip_ctx_group_end"); }

DeviceTimer* HipBackend::createDeviceTimer(const std::string& name) { return new HipDeviceTimer(ctx, name); }

void HipBackend::postRun(RenderStagesArray&, bool dumpOutput, const std::string& folder) {
    (void) folder;
    (void) dumpOutput; // dumps are written by the layers' render passes right after they run (GenericModelLayer::run)
}

// Hand the stage DAG to snnhip_graph_fuse (the library's single fusion pass: residual Conv2D -> Add pairs and linear runs, rules A-F of
// snnhip_chain_plan_create) and install the fused plans it returns.  A fused plan sits at the LAST stage of its group and reads the
// group's external inputs; the other stages of the group keep their textures un-produced and their passes skipped.
void HipBackend::finalizeStages(RenderStagesArray& stages, bool dumpOutputs, bool fuseChains) {
    if (dumpOutputs || !fuseChains) return; // dumps need every intermediate tensor
    auto passOf = [&](size_t i) -> HipRenderPass* {
        auto* ml = static_cast<GenericModelLayer*>(stages[i].layer->modelLayer);
        if (!ml || stages[i].layer->isInputLayer || stages[i].backend != Backend::Backend_GPU || ml->getRenderPasses().size() != 1) return nullptr;
        return dynamic_cast<HipRenderPass*>(ml->getRenderPasses()[0].get());
    };
    const int n = static_cast<int>(stages.size());
    std::vector<snnhip_graph_node> nodes(stages.size());
    std::vector<snnhip_fused_node> fused(stages.size());
    for (size_t i = 0; i < stages.size(); ++i) {
        snnhip_graph_node& nd = nodes[i];
        nd = snnhip_graph_node{};
        HipRenderPass* rp = passOf(i);
        nd.plan = rp ? rp->plan : nullptr;
        nd.keep = (i + 1 == stages.size()) ? 1 : 0; // the model output (the reference binds the last stage's texture, core.cpp:219-227)
        if (stages[i].inputIds.size() > SNNHIP_GRAPH_MAX_INPUTS) nd.plan = nullptr;
        nd.n_inputs = nd.plan ? static_cast<int>(stages[i].inputIds.size()) : 0;
        for (int k = 0; k < nd.n_inputs; ++k) {
            const bool modelInput = stages[i].delayBindMask[static_cast<size_t>(k)] != 0;
            nd.inputs[k] = modelInput ? -(stages[i].inputIds[static_cast<size_t>(k)] + 1) : stages[i].inputIds[static_cast<size_t>(k)];
        }
    }
    // an opaque stage (CPU layer, multi-pass layer) still consumes its producers: they must stay materialised
    for (size_t i = 0; i < stages.size(); ++i)
        if (!nodes[i].plan)
            for (size_t k = 0; k < stages[i].inputIds.size(); ++k)
                if (!stages[i].delayBindMask[k] && stages[i].inputIds[k] >= 0) nodes[static_cast<size_t>(stages[i].inputIds[k])].keep = 1;
    hipChk(snnhip_graph_fuse(ctx, nodes.data(), n, fused.data()), "snnhip_graph_fuse");
    // which texture carries input `id` of a fused plan: the stage output, or (model inputs, bound at run()) the input slot of the stage that named it
    auto textureOf = [&](size_t groupLast, int id) -> ImageTexture* {
        if (id >= 0) return &stages[static_cast<size_t>(id)].stageOutputs[0];
        for (size_t s = 0; s <= groupLast; ++s)
            for (size_t k = 0; k < stages[s].inputIds.size(); ++k)
                if (stages[s].delayBindMask[k] && -(stages[s].inputIds[k] + 1) == id) return &stages[s].stageInputs[k];
        SNN_RIP("finalizeStages: model input %d is not bound by any stage", -id - 1);
        return nullptr;
    };
    for (size_t i = 0; i < stages.size(); ++i) {
        if (!nodes[i].plan) continue;
        HipRenderPass* rp = passOf(i);
        if (!fused[i].plan) { // folded into a later stage's plan
            rp->skip = true;
            stages[i].fusedAway = true;
            continue;
        }
        bool rewired = fused[i].n_inputs != nodes[i].n_inputs;
        for (int k = 0; k < fused[i].n_inputs && !rewired; ++k) rewired = fused[i].inputs[k] != nodes[i].inputs[k];
        if (!fused[i].owned && !rewired) continue;
        if (fused[i].owned) chainPlans.push_back(fused[i].plan); // a re-wired own plan stays owned by the pass kept in replacedPasses
        auto* ml = static_cast<GenericModelLayer*>(stages[i].layer->modelLayer);
        replacedPasses.push_back(ml->getRenderPasses()[0]); // its plan may still be launched by the fused one (unfused steps of a chain)
        auto np = std::make_shared<HipRenderPass>(fused[i].plan, textureOf(i, fused[i].inputs[0]), rp->output, rp->name + " (+fused)", false);
        for (int k = 1; k < fused[i].n_inputs; ++k) np->extraInputs.push_back(textureOf(i, fused[i].inputs[k]));
        ml->getRenderPasses()[0] = np;
        char buf[512];
        snnhip_plan_describe(fused[i].plan, buf, sizeof(buf));
        SNN_LOGI("stage %zu runs a fused plan: %s", i, buf);
    }
    // ---- independent neighbours: launching stage i and the launching stage k in front of it both read tensors that exist before k runs (i does not
    // consume k's output) -- the two branches of a residual block's entry.  Mark i: run() issues it on the side stream next to k.  One pair at a
    // time (k itself must not be the side stage of another pair).  Opt-in (SNN_BRANCH_OVERLAP=1): measured on ResNet-18 b32 the 1x1 stride-2
    // downsample (19 us alone) and the split-K 3x3 stride-2 convolution (53 us alone) do run concurrently (137 / 181 us summed over the three pairs
    // instead of 56 / 160) but each already fills the chip: 1.104 ms per step with the overlap, 1.101 ms without.
    // Round 6: when BOTH stages of such a pair run plans the library can put into one grid (snnhip_plan_groupable: conv2d_ksplit's 3x3 stride-2 convolution
    // and the 1x1 stride-2 downsample beside it), run() brackets them with a launch group instead -- one launch, the short blocks of the second behind the
    // first's; the default (SNN_STAGE_GROUPS=0: off).
    const char* overlap = getenv("SNN_BRANCH_OVERLAP");
    const char* groupsEnv = getenv("SNN_STAGE_GROUPS");
    const bool wantOverlap = overlap && atoi(overlap) != 0, wantGroups = !(groupsEnv && atoi(groupsEnv) == 0);
    if (!wantOverlap && !wantGroups) return;
    int prev = -1;
    for (size_t i = 0; i < stages.size(); ++i) {
        HipRenderPass* rp = passOf(i);
        if (!rp || rp->skip || !nodes[i].plan) {
            if (!stages[i].layer->isInputLayer && !(rp && rp->skip)) prev = -1; // an opaque stage: nothing pairs across it
            continue;
        }
        const int nIn = fused[i].plan ? fused[i].n_inputs : nodes[i].n_inputs;
        const int* ins = fused[i].plan ? fused[i].inputs : nodes[i].inputs;
        bool independent = prev >= 0 && !stages[static_cast<size_t>(prev)].sideOfPrevious && !stages[static_cast<size_t>(prev)].groupWithPrevious;
        for (int k = 0; k < nIn && independent; ++k) independent = ins[k] != prev;
        // the previous stage's fused group must not contain a producer of i either (a fused plan sits at the LAST stage of its group)
        for (int k = 0; k < nIn && independent; ++k)
            if (ins[k] >= 0 && ins[k] < prev && stages[static_cast<size_t>(ins[k])].fusedAway) independent = false;
        if (independent && i + 1 < stages.size()) { // (never the model's last stage: its output is bound to the caller)
            const snnhip_plan* pi = fused[i].plan ? fused[i].plan : nodes[i].plan;
            const snnhip_plan* pk = fused[static_cast<size_t>(prev)].plan ? fused[static_cast<size_t>(prev)].plan : nodes[static_cast<size_t>(prev)].plan;
            if (wantGroups && snnhip_plan_groupable(pi) && snnhip_plan_groupable(pk)) {
                stages[i].groupWithPrevious = true;
                SNN_LOGI("stage %zu (%s) and stage %d (%s) run in one launch group", i, stages[i].layer->name.c_str(), prev, stages[static_cast<size_t>(prev)].layer->name.c_str());
            } else if (wantOverlap) {
                stages[i].sideOfPrevious = true;
                SNN_LOGI("stage %zu (%s) runs beside stage %d (%s) on the side stream", i, stages[i].layer->name.c_str(), prev, stages[static_cast<size_t>(prev)].layer->name.c_str());
            }
        }
        prev = static_cast<int>(i);
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
