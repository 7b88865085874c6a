// core.cpp -- MixedInferenceCore: stage array, texture wiring, the per-inference host loop
// (reference core/src/ic2/core.cpp:80-95 create, :97-245 run, :247-289 mapDeviceBackend, :294-410 init, :412-451 stats).
#include <sys/stat.h>

#include <sstream>

#include "../../include/snnhip.h"
#include "ic2/backend.h"
#include "ic2/dp.h"
#include "ic2/genericlayer.h"
#include "snn/core.h"

using namespace snn;

MixedInferenceCore::MixedInferenceCore(GpuContext* context_) : context(context_) {}

std::unique_ptr<MixedInferenceCore> MixedInferenceCore::create(GpuContext* context, const CreationParameters& cp) {
    std::unique_ptr<MixedInferenceCore> p(new MixedInferenceCore(context));
    p->init(cp);
    return p;
}

std::unique_ptr<MixedInferenceCore> MixedInferenceCore::create(GpuContext* context, const std::string& modelFileName, const dp::ShaderGenOptions& options,
                                                               bool dumpOutputs) { // core.cpp:86-95
    auto layers = dp::loadFromJsonModel(modelFileName, false, options.mrtMode, options.weightMode, options.preferrHalfPrecision);
    CreationParameters cp;
    static_cast<InferenceGraph&>(cp) = dp::generateInferenceGraph(layers, options);
    cp.dumpOutputs = dumpOutputs;
    cp.fuseChains = options.fuseChains;
    return create(context, cp);
}

// core.cpp:247-289 with GPU_HIP treated like GPU_VK
static std::pair<Backend, Transition> mapDeviceBackend(InferenceGraph::LayerExecutionType prev, InferenceGraph::LayerExecutionType curr) {
    using T = InferenceGraph::LayerExecutionType;
    const bool currGpu = curr != T::CPU && curr != T::NOT_DEFINED;
    const bool prevGpu = prev != T::CPU && prev != T::NOT_DEFINED;
    Backend b = currGpu ? Backend::Backend_GPU : (curr == T::CPU ? Backend::Backend_CPU : Backend::NOT_DEFINED);
    Transition t = Transition::NOT_DEFINED;
    if (prev != T::NOT_DEFINED && prevGpu != currGpu) t = currGpu ? Transition::Backend_CPU_GPU : Transition::Backend_GPU_CPU;
    return {b, t};
}

bool MixedInferenceCore::init(const CreationParameters& cp_) {
    if (cp_.dumpOutputs) mkdir(outputDir(), 0755); // createDirIfNotExists(OUTPUT_DIR), core.cpp:295-297
    SNN_ASSERT(cp_.inputsDesc.size() > 0);
    this->cp = cp_;
    backend = dp::BackendBuilder::build(context, cp);
    if (cp.profiling) gpuRunTime = backend->createDeviceTimer("IC2 Total GPU runtime");

    stages.reserve(cp.layers.size());
    for (size_t i = 0; i < cp.layers.size(); ++i) stages.emplace_back(context);

    InferenceGraph::LayerExecutionType preDev = InferenceGraph::LayerExecutionType::NOT_DEFINED;
    for (size_t i = 0; i < cp.layers.size(); i++) { // core.cpp:318-330
        if (cp.layers[i]->flattenLayer) bindOutput = false;
        auto bt = mapDeviceBackend(preDev, cp.layers[i]->layerLoc);
        stages[i].backend = bt.first;
        stages[i].transition = bt.second;
        stages[i].delayBindMask.resize(cp.layers[i]->inputRefs.size(), 0);
        preDev = cp.layers[i]->layerLoc;
    }

    for (size_t i = 0; i < stages.size(); ++i) { // core.cpp:332-406
        InferenceGraph::Layer& layer = *cp.layers[i];
        RenderStage& stage = stages[i];
        stage.layer = cp.layers[i];
        // the only CPU stage of the HIP flavour is the YOLO head, a CPU layer in the reference as well (yololayer.h:38); Dense / Flatten run on the GPU
        const bool cpuStage = stage.backend != Backend::Backend_GPU;
        stage.stageInputs.allocate(layer.inputRefs.size());
        stage.stageOutputs.allocate(1);
        if (stage.layer->isInputLayer) {
            if (cp.profiling) stage.timer.reset(backend->createDeviceTimer(layer.name));
            continue; // nothing to create or bind for the input layer
        }
        for (size_t j = 0; j < layer.inputRefs.size(); ++j) {
            const auto& ref = layer.inputRefs[j];
            if (ref.isStageOutput) {
                SNN_ASSERT(ref.index < static_cast<int>(i));
                stage.stageInputs[j].attach(&stages[static_cast<size_t>(ref.index)].stageOutputs[0]);
                stage.inputIds.push_back(ref.index);
            } else { // delay binding: the model input image arrives with run() (core.cpp:365-369)
                const auto inputIdx = stages[static_cast<size_t>(ref.index)].layer->inputIndex;
                stage.delayBindMask[j] = 1;
                stage.inputIds.push_back(static_cast<int>(inputIdx));
            }
        }
        std::array<uint32_t, 4> dims{layer.outputDesc.width, layer.outputDesc.height, layer.outputDesc.depth, layer.outputDesc.batch}; // the reference: {W,H,D,1}
        if (!cpuStage) { // a CPU stage hands its result over as a host matrix (ImageTexture::setOutputMat), no device tensor
            stage.stageOutputs[0].resetTexture(dims, layer.outputDesc.format, layer.name, layer.outputDesc.channels); // core.cpp:371-372
            layer.initFunPtr(backend, stage.stageInputs, stage.stageOutputs);                                        // core.cpp:374
        }
        if (cp.profiling) stage.timer.reset(backend->createDeviceTimer(layer.name));
    }
    backend->finalizeStages(stages, cp.dumpOutputs, cp.fuseChains);
    graphUsable = cp.captureGraph && !cp.dumpOutputs && !cp.profiling;
    for (auto& s : stages)
        if (!s.layer->isInputLayer && s.backend != Backend::Backend_GPU) graphUsable = false;
    if (graphUsable) {
        // A recorded graph pays off when an inference is MANY launches (ResNet-18 23, MobileNetV2 41, Candy 88 after fusion).  For a handful it costs:
        // consecutive hipGraphLaunch calls leave ~5 us between graphs on the stream where plain kernel launches queue back to back (measured on the
        // fused ESPCN, 2 launches of 86 + 34 us: 125.0 us per inference replayed, 120 launched directly) -- below the threshold run() just launches.
        int launches = 0;
        for (auto& s : stages) {
            auto* ml = static_cast<dp::GenericModelLayer*>(s.layer->modelLayer);
            if (!ml || s.layer->isInputLayer) continue;
            for (auto& rp : ml->getRenderPasses())
                if (auto* hp = dynamic_cast<dp::HipRenderPass*>(rp.get()))
                    if (!hp->skip && hp->plan) launches += snnhip_plan_num_steps(hp->plan);
        }
        const char* minL = getenv("SNN_GRAPH_MIN_LAUNCHES");
        if (launches < (minL ? atoi(minL) : 6)) graphUsable = false;
    }
    return true;
}

void MixedInferenceCore::run(RunParameters& rp) { // core.cpp:97-245
    SNN_ASSERT(rp.inputImages && rp.inputImages->size() > 0);
    if (rp.inputImages->size() != cp.inputsDesc.size()) {
        SNN_LOGE("Wrong input texture count %zu <-> %zu", rp.inputImages->size(), cp.inputsDesc.size());
        return;
    }
    backend->prepareRun(rp, stages, bindOutput, static_cast<uint32_t>(stages.size() - 1));
    cpuRunTime.start();
    if (gpuRunTime) gpuRunTime->start();
    bool replayed = false, recordingNow = false;
    if (graphUsable && !replaySuspended) {
        // The captured kernels hold DEVICE pointers: key the recording on (device address, dims, dtype) of every model input, not on
        // the host-side tensor handle (an input texture that re-creates its tensor may get the same handle address back with another
        // buffer behind it, and the other way round).
        std::vector<InputKey> ins;
        for (size_t k = 0; k < rp.inputImages->size(); ++k) {
            const snnhip_tensor* t = (*rp.inputImages)[k].tensor();
            InputKey key{};
            if (t) {
                key.data = snnhip_tensor_data(t);
                snnhip_tensor_dims(t, key.dims);
                key.dtype = snnhip_tensor_dtype(t);
            }
            ins.push_back(key);
        }
        if (ins == recordedInputs && backend->replay()) {
            replayed = true; // same input buffers as when the launch sequence was recorded: one host call
        } else {
            recordedInputs = ins;
            recordingNow = backend->beginRecord();
        }
    }
    auto runStage = [&](RenderStage& s) {
        for (size_t n = 0; n < s.delayBindMask.size(); ++n) {
            if (s.delayBindMask[n] > 0) s.stageInputs[n].attach(&(*rp.inputImages)[static_cast<size_t>(s.inputIds[n])]); // core.cpp:127-133
        }
        if (s.timer) s.timer->start();
        backend->prepareStage(rp, s);
        if (s.backend == Backend::Backend_GPU) {
            s.layer->runFunPtr(backend, cp.dumpOutputs);
        } else { // GPU -> CPU transition: wait for the producers, then the layer reads them back itself (core.cpp:141-199)
            backend->sync();
            s.layer->imageTextureFunPtr(s.stageInputs, s.stageOutputs);
        }
        if (s.timer) s.timer->stop();
    };
    const bool sideBySide = !cp.profiling && !cp.dumpOutputs; // per-stage timers and dumps want one stream and the reference's order
    for (size_t i = 0; i < stages.size() && !replayed; i++) {
        auto& s = stages[i];
        if (s.layer->isInputLayer) continue;
        if (s.fusedAway) { // its pass is skipped, but a model input may be bound through it (the fused plan of a later stage reads that slot)
            runStage(s);
            continue;
        }
        // HIP extension: the next launching stage is independent of this one (HipBackend::finalizeStages) -> it goes to the side stream first, this one
        // runs on the main stream beside it, and the main stream waits for both before the stage that consumes them
        size_t j = i + 1;
        while (j < stages.size() && (stages[j].layer->isInputLayer || stages[j].fusedAway)) ++j;
        if (sideBySide && j < stages.size() && stages[j].groupWithPrevious && backend->groupBegin()) { // one launch for both (HipBackend::finalizeStages)
            runStage(s);
            for (size_t m = i + 1; m < j; ++m)
                if (!stages[m].layer->isInputLayer) runStage(stages[m]); // (fused-away stages in between: input binding only, nothing is launched)
            runStage(stages[j]);
            backend->groupEnd();
            i = j;
            continue;
        }
        if (sideBySide && j < stages.size() && stages[j].sideOfPrevious && backend->forkSide()) {
            for (size_t m = i + 1; m < j; ++m)
                if (!stages[m].layer->isInputLayer) runStage(stages[m]); // (fused-away stages in between: input binding only, nothing is launched)
            runStage(stages[j]);
            backend->backToMain();
            runStage(s);
            backend->joinSide();
            i = j;
            continue;
        }
        runStage(s);
    }
    if (recordingNow) { // the loop above only recorded: submit it now
        if (!backend->endRecord() || !backend->replay()) SNN_RIP("hipGraph capture of the inference failed: %s", snnhip_last_error());
    }
    if (gpuRunTime) gpuRunTime->stop();
    bool hostStage = false;
    for (auto& s : stages)
        if (!s.layer->isInputLayer && s.backend != Backend::Backend_GPU) hostStage = true;
    const bool defer = rp.deferSync && !cp.profiling && !cp.dumpOutputs && !hostStage && rp.modelOutput.modelType != ModelType::CLASSIFICATION &&
                       rp.modelOutput.modelType != ModelType::DETECTION;
    if (!defer) backend->sync(); // the only GPU wait of an inference (core.cpp:203)
    backend->postRun(stages, cp.dumpOutputs, outputDir());
    if (!defer) {
        for (auto& s : stages)
            if (s.timer && !s.layer->isInputLayer) s.timer->getTime();
        if (gpuRunTime) gpuRunTime->getTime();
    }
    cpuRunTime.stop();
    if (rp.outputImages && rp.outputImages->size() > 0 && bindOutput) {
        // the reference binds the last stage's texture to the caller's output image (Android path); here the caller's
        // texture simply aliases the last stage output
        (*rp.outputImages)[0].attach(&stages.back().stageOutputs[0]);
    }
    // core.cpp:228-237.  The classifier head (Dense + softmax) is a GPU stage here, so the arg-max is taken on the device tensor;
    // like the reference the reported class is 1-based (0 = none)
    if (rp.modelOutput.modelType == ModelType::CLASSIFICATION && stages.back().backend == Backend::Backend_GPU && stages.back().stageOutputs[0].tensor()) {
        int idx = -1;
        if (snnhip_tensor_argmax(stages.back().stageOutputs[0].tensor(), 0, &idx) != SNNHIP_OK) SNN_RIP("snnhip_tensor_argmax: %s", snnhip_last_error());
        rp.modelOutput.classifierOutput = idx + 1;
        SNN_LOGD("Classifier output: %d", rp.modelOutput.classifierOutput);
    } else if (rp.modelOutput.modelType == ModelType::DETECTION && stages.back().backend == Backend::Backend_CPU) {
        rp.modelOutput.detectionOutput = stages.back().stageOutputs[0].getOutputMat();
    }
    backend->cleanupRun();
}

bool MixedInferenceCore::sync() { return backend->sync(); }

void MixedInferenceCore::writeTimeStat(std::map<std::string, std::vector<double>>& timeArray) { // core.cpp:437-442
    if (gpuRunTime) timeArray[gpuRunTime->getName()].push_back(gpuRunTime->duration() / 1000000.0);
    for (auto& s : stages)
        if (s.timer) timeArray[s.timer->getName()].push_back(s.timer->duration() / 1000000.0);
}

std::string MixedInferenceCore::describe() const {
    std::ostringstream ss;
    for (size_t i = 0; i < stages.size(); ++i) {
        ss << "[" << i << "] " << stages[i].layer->name;
        if (stages[i].fusedAway) ss << "  (fused into a later stage's plan)";
        ss << "\n";
    }
    return ss.str();
}

MixedInferenceCore::~MixedInferenceCore() {
    stages.clear();
    cp.layers.clear();
    delete backend;
    delete gpuRunTime;
}
