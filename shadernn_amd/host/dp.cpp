// dp.cpp -- JSON model -> layer DAG -> InferenceGraph (reference core/src/ic2/dp.cpp:115-167, 389-640).
#include <map>
#include <queue>
#include <sstream>
#include <unordered_map>

#include "ic2/dp.h"
#include "ic2/layerFactory.h"

using namespace snn;
using namespace snn::dp;

InferenceModel snn::dp::loadFromJsonModel(const std::string& fileName, bool useVulkan, const MRTMode& mrtMode, const WeightAccessMethod& weightMode,
                                          bool preferHp) {
    InferenceModel layers;
    ModelParser parser({fileName, preferHp, mrtMode, weightMode});
    const int32_t layerCount = parser.getLayerCount();
    initLayerRegisty();
    const size_t slash = fileName.find_last_of('/');
    const std::string shortName = slash == std::string::npos ? fileName : fileName.substr(slash + 1);
    for (int i = 0; i < layerCount; i++) {
        SNN_ASSERT(parser.getNumInbound(i) == static_cast<int>(parser.getInboundLayerId(i).size()));
        const std::string layerName = parser.getLayerName(i);
        layers.emplace_back(std::shared_ptr<GenericModelLayer>(createLayerInstance(layerName, parser, i, useVulkan)));
        layers.back()->setName(formatString("%s layer [%02d] %s", shortName.c_str(), i, layerName.c_str())); // dp.cpp:134
    }
    if (layers.empty()) {
        SNN_LOGE("head layer not found.");
        return {};
    }
    for (int i = 0; i < layerCount; i++) { // build layer connections (dp.cpp:144-156)
        for (int ii : parser.getInboundLayerId(i)) {
            layers[static_cast<size_t>(i)]->prevLayers.push_back(layers[static_cast<size_t>(ii)]);
            layers[static_cast<size_t>(ii)]->nextLayers.push_back(layers[static_cast<size_t>(i)]);
        }
    }
    return layers;
}

// Kahn's algorithm over all layers (dp.cpp:389-429)
static std::vector<std::shared_ptr<GenericModelLayer>> topologicalSort2(const std::vector<std::shared_ptr<GenericModelLayer>>& layers) {
    std::vector<std::shared_ptr<GenericModelLayer>> sorted;
    std::unordered_map<GenericModelLayer*, size_t> inDegree;
    for (auto& n : layers)
        for (auto& nx : n->nextLayers) inDegree[nx.get()]++;
    std::queue<std::shared_ptr<GenericModelLayer>> processing;
    for (auto& n : layers)
        if (inDegree[n.get()] == 0) processing.push(n);
    while (!processing.empty()) {
        auto n = processing.front();
        processing.pop();
        sorted.push_back(n);
        for (auto& nx : n->nextLayers)
            if (--inDegree[nx.get()] == 0) processing.push(nx);
    }
    if (sorted.size() != layers.size()) SNN_LOGW("There exists a cycle in the graph !");
    return sorted;
}

InferenceGraph snn::dp::generateInferenceGraph(std::vector<std::shared_ptr<GenericModelLayer>>& layers, const ShaderGenOptions& options) {
    auto modelLayers = topologicalSort2(layers);
    InferenceGraph graph;
    graph.mrtMode = options.mrtMode;
    graph.weightMode = options.weightMode;
    std::map<GenericModelLayer*, size_t> s2i;
    uint32_t inputLayers = 0;
    const ColorFormat fmt = options.preferrHalfPrecision ? ColorFormat::RGBA16F : ColorFormat::RGBA32F;

    for (auto& modelLayer : modelLayers) { // dp.cpp:446-478
        graph.layers.emplace_back(new InferenceGraph::Layer);
        auto* igLayer = graph.layers.back().get();
        igLayer->imageTextureFunPtr = [modelLayer](ImageTextureArray& in, ImageTextureArray& out) { modelLayer->computeImageTexture(in, out); };
        igLayer->initFunPtr = [modelLayer](DeviceBackend* backend, ImageTextureArray& in, ImageTextureArray& out) { modelLayer->init(backend, in, out); };
        igLayer->runFunPtr = [modelLayer](DeviceBackend* backend, bool dumpOutputs) { modelLayer->run(backend, dumpOutputs); };
        igLayer->modelLayer = modelLayer.get();
        igLayer->layerLoc = modelLayer->getLayerExecutionType();
        igLayer->name = modelLayer->getName();
        igLayer->isInputLayer = modelLayer->isInputLayer();
        if (modelLayer->isInputLayer()) {
            inputLayers++;
            igLayer->inputIndex = modelLayer->getInputIndex();
        }
    }

    std::ostringstream modelFormat;
    modelFormat << "================================================================\n";
    modelFormat << "|  Layer ID  |              Name                 | Output Dims |\n";
    modelFormat << "================================================================\n";
    for (size_t i = 0; i < graph.layers.size(); ++i) { // dp.cpp:491-632
        auto* igLayer = graph.layers[i].get();
        auto& modelLayer = modelLayers[i];
        modelLayer->setMRTMode(options.mrtMode);
        modelLayer->setWeightAccessMode(options.weightMode);
        s2i[modelLayer.get()] = i;
        uint32_t inputWidth = 0, inputHeight = 0, width = 0, height = 0, depth = 0;
        auto inputDesc = [&](uint32_t idx) {
            SNN_CHK(idx < options.desiredInput.size());
            const auto& di = options.desiredInput[idx];
            // the reference passes channels = 4*depth for model inputs (dp.cpp:506-508); the true count, when the caller
            // gives one, lets the NHWC tensors carry exactly C channels
            return InferenceGraph::IODesc{fmt, di.width, di.height, di.depth, di.channels ? di.channels : 4 * di.depth, options.batch};
        };
        if (!modelLayer->prevLayers.empty()) {
            for (auto& prev : modelLayer->prevLayers) {
                InferenceGraph::LayerRef ref;
                ref.index = static_cast<int>(s2i[prev.get()]);
                InferenceGraph::IODesc imageInput;
                if (prev->isInputLayer()) {
                    ref.isStageOutput = false;
                    imageInput = inputDesc(prev->getInputIndex());
                    if (prev->getDesc().numOutputPlanes) imageInput.channels = prev->getDesc().numOutputPlanes; // InputLayer "outputPlanes" = true channels
                } else {
                    ref.isStageOutput = true;
                    imageInput = graph.layers[static_cast<size_t>(ref.index)]->outputDesc;
                }
                modelLayer->addInputDim(imageInput);
                inputWidth = std::max(inputWidth, imageInput.width);
                inputHeight = std::max(inputHeight, imageInput.height);
                igLayer->inputRefs.push_back(ref);
            }
            modelLayer->getOutputDims(width, height, depth);
        } else { // input layers
            InferenceGraph::LayerRef ref;
            ref.isStageOutput = false;
            ref.index = -1;
            auto imageInput = inputDesc(modelLayer->getInputIndex());
            modelLayer->addInputDim(imageInput);
            inputWidth = imageInput.width;
            inputHeight = imageInput.height;
            modelLayer->getOutputDims(width, height, depth);
            igLayer->inputRefs.push_back(ref);
        }
        const std::string& nm = modelLayer->getName();
        const size_t br = nm.find('[');
        std::string layerName = br == std::string::npos ? nm : nm.substr(br);
        if (layerName.size() > 34) layerName = layerName.substr(0, 31) + "...";
        const std::string dims = std::to_string(width) + " x " + std::to_string(height) + " x " + std::to_string(depth);
        modelFormat << "| " << i << std::string(i > 9 ? 9 : 10, ' ') << "| " << layerName << std::string(34 - layerName.size(), ' ') << "| " << dims
                    << std::string(dims.size() < 12 ? 12 - dims.size() : 0, ' ') << "|\n";

        if (igLayer->layerLoc != InferenceGraph::LayerExecutionType::CPU) {
            GenericModelLayer::LayerGenOptions opt;
            static_cast<ShaderGenOptions&>(opt) = options;
            if (!opt.desiredInput.empty()) {
                opt.desiredInput[0].width = inputWidth;
                opt.desiredInput[0].height = inputHeight;
            }
            opt.desiredOutputWidth = width;
            opt.desiredOutputHeight = height;
            opt.isFirstLayer = (i == inputLayers);
            opt.isLastLayer = (i == graph.layers.size() - 1);
            if (modelLayer->isInputLayer()) {
                modelLayer->setLayerExecutionType(InferenceGraph::LayerExecutionType::GPU_HIP);
            } else {
                modelLayer->createInferencePasses(opt);
            }
            igLayer->layerLoc = modelLayer->getLayerExecutionType();
            igLayer->outputDesc = {fmt, width, height, static_cast<uint32_t>(DIV_4_ROUND_UP(modelLayer->getDesc().numOutputPlanes)),
                                   modelLayer->getDesc().numOutputPlanes, options.batch}; // dp.cpp:328-332
            if (modelLayer->isInputLayer()) igLayer->outputDesc = inputDesc(modelLayer->getInputIndex());
            if (dynamic_cast<DenseLayer*>(modelLayer.get())) {
                // Dense output is a units x 1 x 1 single-channel image in the reference (denselayer.cpp:40-55, CPU-stage
                // descriptor dp.cpp:365-367); it runs on the GPU here but keeps that shape
                igLayer->outputDesc = {fmt, width, 1, 1, 1, options.batch};
                igLayer->flattenLayer = true;
            }
            if (dynamic_cast<FlattenLayer*>(modelLayer.get())) { // W*H*C x 1 x 1, one channel (flattenlayer.cpp:48-62)
                igLayer->outputDesc = {fmt, width, 1, 1, 1, options.batch};
                igLayer->flattenLayer = true;
            }
            SNN_ASSERT(igLayer->outputDesc.width > 0 && igLayer->outputDesc.height > 0);
        } else {
            if (i == 0) SNN_RIP("CPU layer currently cannot cannot be the 1-st layer in the graph !");
            if (options.batch != 1) SNN_RIP("CPU layer %s: CPU stages (the YOLO head) take one image per inference, batch = %u", modelLayer->getName().c_str(), options.batch);
            igLayer->outputDesc = {fmt, width, height, depth, modelLayer->getDesc().numOutputPlanes};
        }
        modelFormat << "----------------------------------------------------------------\n";
    }
    graph.inputsDesc = options.desiredInput;
    modelFormat << "================================================================\n";
    SNN_LOGI("\n%s", modelFormat.str().c_str());
    return graph;
}

InferenceGraph snn::dp::generateInferenceGraph(const std::shared_ptr<GenericModelLayer> firstLayer, const ShaderGenOptions& options) {
    // collect everything reachable from the head (the reference walks nextLayers with a BFS, dp.cpp:33-55)
    std::vector<std::shared_ptr<GenericModelLayer>> all;
    std::queue<std::shared_ptr<GenericModelLayer>> q;
    std::map<GenericModelLayer*, bool> seen;
    q.push(firstLayer);
    seen[firstLayer.get()] = true;
    while (!q.empty()) {
        auto n = q.front();
        q.pop();
        all.push_back(n);
        for (auto& nx : n->nextLayers)
            if (!seen[nx.get()]) {
                seen[nx.get()] = true;
                q.push(nx);
            }
    }
    return generateInferenceGraph(all, options);
}
