// dp.cpp -- JSON model -> layer DAG -> InferenceGraph.  The interface (ic2/dp.h) and the observable results -- stage order, names, descriptors, the layer
// table in the log -- are the reference's (core/src/ic2/dp.cpp:115-167 model loading, :389-429 ordering, :432-640 graph generation); the construction is
// this repository's: layers are numbered once, the order is computed over index arrays, and a small builder owns the per-stage bookkeeping.
#include <algorithm>
#include <cstdio>
#include <string>
#include <unordered_map>
#include <vector>

#include "ic2/dp.h"
#include "ic2/layerFactory.h"

using namespace snn;
using namespace snn::dp;

namespace {

typedef std::shared_ptr<GenericModelLayer> LayerPtr;

std::string baseName(const std::string& path) {
    const size_t cut = path.find_last_of('/');
    return cut == std::string::npos ? path : path.substr(cut + 1);
}

// Stage order: breadth-first by resolved dependencies, ties in the order the layers were handed in (the order the reference's queue-based pass produces,
// dp.cpp:389-429 -- stage indices name dump files and timers, so the order is part of the contract).  Index arrays instead of pointer-keyed maps.
std::vector<LayerPtr> stageOrder(const std::vector<LayerPtr>& layers) {
    const size_t count = layers.size();
    std::unordered_map<const GenericModelLayer*, size_t> slot;
    slot.reserve(count);
    for (size_t i = 0; i < count; ++i) slot.emplace(layers[i].get(), i);
    std::vector<size_t> pending(count, 0); // unresolved producers per layer
    for (const LayerPtr& layer : layers)
        for (const LayerPtr& consumer : layer->nextLayers) {
            auto it = slot.find(consumer.get());
            if (it != slot.end()) pending[it->second]++;
        }
    std::vector<size_t> ready; // a FIFO that is never popped: `head` walks it
    ready.reserve(count);
    for (size_t i = 0; i < count; ++i)
        if (pending[i] == 0) ready.push_back(i);
    for (size_t head = 0; head < ready.size(); ++head)
        for (const LayerPtr& consumer : layers[ready[head]]->nextLayers) {
            auto it = slot.find(consumer.get());
            if (it != slot.end() && --pending[it->second] == 0) ready.push_back(it->second);
        }
    if (ready.size() != count) { // a malformed model: fail here, by name, not later inside the builder (the C ABI has no way to carry an exception out)
        std::string stuck;
        for (size_t i = 0; i < count && stuck.size() < 200; ++i)
            if (pending[i] != 0) stuck += (stuck.empty() ? "" : ", ") + layers[i]->getName();
        SNN_RIP("the layer graph has a cycle: %zu of %zu layers can be ordered; waiting for a producer that never runs: %s", ready.size(), count, stuck.c_str());
    }
    std::vector<LayerPtr> ordered;
    ordered.reserve(ready.size());
    for (size_t i : ready) ordered.push_back(layers[i]);
    return ordered;
}

// The layer table of the log (same columns as the reference prints, dp.cpp:482-636)
class LayerTable {
public:
    LayerTable() {
        rule('=');
        text += "|  Layer ID  |              Name                 | Output Dims |\n";
        rule('=');
    }
    void row(size_t id, const std::string& fullName, uint32_t w, uint32_t h, uint32_t d) {
        const size_t bracket = fullName.find('[');
        std::string name = bracket == std::string::npos ? fullName : fullName.substr(bracket);
        if (name.size() > 34) name = name.substr(0, 31) + "...";
        char dims[64];
        snprintf(dims, sizeof(dims), "%u x %u x %u", w, h, d);
        char line[160];
        snprintf(line, sizeof(line), "| %-*zu| %-34s| %-12s|\n", 11, id, name.c_str(), dims);
        text += line;
        rule('-');
    }
    const std::string& finish() {
        rule('=');
        return text;
    }

private:
    void rule(char c) { text += std::string(64, c) + "\n"; }
    std::string text;
};

// One pass over the ordered layers: stage records, input wiring, output descriptors, inference passes
class GraphBuilder {
public:
    GraphBuilder(const ShaderGenOptions& o) : options(o), format(o.preferrHalfPrecision ? ColorFormat::RGBA16F : ColorFormat::RGBA32F) {
        graph.mrtMode = o.mrtMode;
        graph.weightMode = o.weightMode;
    }

    InferenceGraph build(const std::vector<LayerPtr>& ordered) {
        for (const LayerPtr& layer : ordered) addStage(layer);
        LayerTable table;
        for (size_t i = 0; i < ordered.size(); ++i) {
            uint32_t w = 0, h = 0, d = 0;
            describeStage(i, ordered[i], w, h, d);
            table.row(i, ordered[i]->getName(), w, h, d);
        }
        graph.inputsDesc = options.desiredInput;
        SNN_LOGI("\n%s", table.finish().c_str());
        return std::move(graph);
    }

private:
    const ShaderGenOptions& options;
    const ColorFormat format;
    InferenceGraph graph;
    std::unordered_map<const GenericModelLayer*, int> stageOf;
    uint32_t inputStages = 0;

    // the descriptor of model input `idx`.  The reference passes channels = 4 * depth (dp.cpp:506-508); the true count, when the caller gives one, lets
    // the NHWC tensors carry exactly C channels
    InferenceGraph::IODesc modelInput(uint32_t idx) const {
        SNN_CHK(idx < options.desiredInput.size());
        const auto& want = options.desiredInput[idx];
        return InferenceGraph::IODesc{format, want.width, want.height, want.depth, want.channels ? want.channels : 4 * want.depth, options.batch};
    }

    void addStage(const LayerPtr& layer) {
        graph.layers.emplace_back(new InferenceGraph::Layer);
        InferenceGraph::Layer& stage = *graph.layers.back();
        stage.modelLayer = layer.get();
        stage.name = layer->getName();
        stage.layerLoc = layer->getLayerExecutionType();
        stage.isInputLayer = layer->isInputLayer();
        if (stage.isInputLayer) {
            stage.inputIndex = layer->getInputIndex();
            ++inputStages;
        }
        stage.imageTextureFunPtr = [layer](ImageTextureArray& in, ImageTextureArray& out) { layer->computeImageTexture(in, out); };
        stage.initFunPtr = [layer](DeviceBackend* backend, ImageTextureArray& in, ImageTextureArray& out) { layer->init(backend, in, out); };
        stage.runFunPtr = [layer](DeviceBackend* backend, bool dumpOutputs) { layer->run(backend, dumpOutputs); };
    }

    // what stage `producer` hands to a consumer: a model input's descriptor (with the InputLayer's "outputPlanes" as the true channel count) or the
    // producing stage's output
    InferenceGraph::IODesc feed(const LayerPtr& producer, InferenceGraph::LayerRef& ref) const {
        const auto at = stageOf.find(producer.get());
        if (at == stageOf.end()) // (an edge into a layer that was not handed in, or that is ordered behind its consumer: stageOrder() rules out the second)
            SNN_RIP("layer %s is read before it has a stage: it is not part of the layer set the graph is generated from", producer->getName().c_str());
        ref.index = at->second;
        ref.isStageOutput = !producer->isInputLayer();
        if (ref.isStageOutput) return graph.layers[static_cast<size_t>(ref.index)]->outputDesc;
        InferenceGraph::IODesc desc = modelInput(producer->getInputIndex());
        if (producer->getDesc().numOutputPlanes) desc.channels = producer->getDesc().numOutputPlanes;
        return desc;
    }

    void describeStage(size_t i, const LayerPtr& layer, uint32_t& width, uint32_t& height, uint32_t& depth) {
        InferenceGraph::Layer& stage = *graph.layers[i];
        layer->setMRTMode(options.mrtMode);
        layer->setWeightAccessMode(options.weightMode);
        stageOf[layer.get()] = static_cast<int>(i);

        uint32_t inW = 0, inH = 0; // the largest input extent: what a GPU stage is generated for
        if (layer->prevLayers.empty()) { // a model input
            InferenceGraph::LayerRef ref;
            ref.index = -1;
            ref.isStageOutput = false;
            const InferenceGraph::IODesc desc = modelInput(layer->getInputIndex());
            layer->addInputDim(desc);
            inW = desc.width;
            inH = desc.height;
            stage.inputRefs.push_back(ref);
        } else {
            for (const LayerPtr& producer : layer->prevLayers) {
                InferenceGraph::LayerRef ref;
                const InferenceGraph::IODesc desc = feed(producer, ref);
                layer->addInputDim(desc);
                inW = std::max(inW, desc.width);
                inH = std::max(inH, desc.height);
                stage.inputRefs.push_back(ref);
            }
        }
        layer->getOutputDims(width, height, depth);

        if (stage.layerLoc == InferenceGraph::LayerExecutionType::CPU) {
            if (i == 0) SNN_RIP("CPU layer currently cannot cannot be the 1-st layer in the graph !");
            if (options.batch != 1)
                SNN_RIP("CPU layer %s: CPU stages (the YOLO head) take one image per inference, batch = %u", layer->getName().c_str(), options.batch);
            stage.outputDesc = {format, width, height, depth, layer->getDesc().numOutputPlanes};
            return;
        }
        describeGpuStage(i, layer, stage, inW, inH, width, height);
    }

    void describeGpuStage(size_t i, const LayerPtr& layer, InferenceGraph::Layer& stage, uint32_t inW, uint32_t inH, uint32_t width, uint32_t height) {
        if (layer->isInputLayer()) {
            layer->setLayerExecutionType(InferenceGraph::LayerExecutionType::GPU_HIP);
        } else {
            GenericModelLayer::LayerGenOptions gen;
            static_cast<ShaderGenOptions&>(gen) = options;
            if (!gen.desiredInput.empty()) {
                gen.desiredInput[0].width = inW;
                gen.desiredInput[0].height = inH;
            }
            gen.desiredOutputWidth = width;
            gen.desiredOutputHeight = height;
            gen.isFirstLayer = i == inputStages;
            gen.isLastLayer = i + 1 == graph.layers.size();
            layer->createInferencePasses(gen);
        }
        stage.layerLoc = layer->getLayerExecutionType();
        const uint32_t planes = layer->getDesc().numOutputPlanes;
        if (layer->isInputLayer()) {
            stage.outputDesc = modelInput(layer->getInputIndex());
        } else if (dynamic_cast<DenseLayer*>(layer.get()) || dynamic_cast<FlattenLayer*>(layer.get())) {
            // Dense: a units x 1 x 1 single-channel image in the reference (denselayer.cpp:40-55, CPU-stage descriptor dp.cpp:365-367) -- it runs on the
            // GPU here but keeps that shape; Flatten: W*H*C x 1 x 1, one channel (flattenlayer.cpp:48-62)
            stage.outputDesc = {format, width, 1, 1, 1, options.batch};
            stage.flattenLayer = true;
        } else {
            stage.outputDesc = {format, width, height, static_cast<uint32_t>(DIV_4_ROUND_UP(planes)), planes, options.batch}; // (dp.cpp:328-332)
        }
        SNN_ASSERT(stage.outputDesc.width > 0 && stage.outputDesc.height > 0);
    }
};

} // namespace

InferenceModel snn::dp::loadFromJsonModel(const std::string& fileName, bool useVulkan, const MRTMode& mrtMode, const WeightAccessMethod& weightMode,
                                          bool preferHp) {
    ModelParser parser({fileName, preferHp, mrtMode, weightMode});
    initLayerRegisty();
    const std::string file = baseName(fileName);
    const int32_t count = parser.getLayerCount();
    InferenceModel model;
    model.reserve(static_cast<size_t>(std::max(count, 0)));
    for (int32_t i = 0; i < count; ++i) {
        SNN_ASSERT(parser.getNumInbound(i) == static_cast<int>(parser.getInboundLayerId(i).size()));
        const std::string kind = parser.getLayerName(i);
        model.emplace_back(createLayerInstance(kind, parser, i, useVulkan));
        model.back()->setName(formatString("%s layer [%02d] %s", file.c_str(), i, kind.c_str())); // (the reference's stage names, dp.cpp:134)
    }
    if (model.empty()) {
        SNN_LOGE("head layer not found.");
        return {};
    }
    for (int32_t i = 0; i < count; ++i) // edges, both directions
        for (int producer : parser.getInboundLayerId(i)) {
            if (producer < 0 || producer >= count) SNN_RIP("%s: layer %d names inbound layer %d, the model has layers 0..%d", file.c_str(), i, producer, count - 1);
            model[static_cast<size_t>(i)]->prevLayers.push_back(model[static_cast<size_t>(producer)]);
            model[static_cast<size_t>(producer)]->nextLayers.push_back(model[static_cast<size_t>(i)]);
        }
    return model;
}

InferenceGraph snn::dp::generateInferenceGraph(std::vector<std::shared_ptr<GenericModelLayer>>& layers, const ShaderGenOptions& options) {
    return GraphBuilder(options).build(stageOrder(layers));
}

InferenceGraph snn::dp::generateInferenceGraph(const std::shared_ptr<GenericModelLayer> firstLayer, const ShaderGenOptions& options) {
    // everything reachable from the head, breadth first (the single-input form of the reference walks nextLayers the same way, dp.cpp:33-55)
    std::vector<LayerPtr> reach{firstLayer};
    std::unordered_map<const GenericModelLayer*, bool> seen{{firstLayer.get(), true}};
    for (size_t head = 0; head < reach.size(); ++head)
        for (const LayerPtr& next : reach[head]->nextLayers)
            if (seen.emplace(next.get(), true).second) reach.push_back(next);
    return generateInferenceGraph(reach, options);
}
