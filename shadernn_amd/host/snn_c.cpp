// snn_c.cpp -- C binding of the host mirror for tests / harnesses (see include/snn_c.h).
#include <cstring>

#include "../../include/snn_c.h"
#include "../../include/snnhip.h"
#include "ic2/backend.h"
#include "ic2/dp.h"
#include "ic2/json.h"
#include "ic2/layerFactory.h"
#include "snn/contextFactory.h"
#include "snn/core.h"

using namespace snn;

struct snn_model {
    GpuContext* context = nullptr;
    std::unique_ptr<MixedInferenceCore> core;
    ImageTextureArray inputs{nullptr};
    ImageTextureArray outputs{nullptr};
    int inW = 0, inH = 0, inC = 0, batch = 1;
    bool half = false;
    SNNModelOutput modelOutput;
};

static dp::ShaderGenOptions makeOptions(int w, int h, int c, bool fuse, bool half = false, int batch = 1) {
    dp::ShaderGenOptions sgo;
    const ColorFormat fmt = half ? ColorFormat::RGBA16F : ColorFormat::RGBA32F;
    InferenceGraph::IODesc in{fmt, static_cast<uint32_t>(w), static_cast<uint32_t>(h), static_cast<uint32_t>(UP_DIV(c, 4)), static_cast<uint32_t>(c)};
    sgo.desiredInput.push_back(in);
    sgo.desiredOutputFormat = fmt;
    sgo.preferrHalfPrecision = half;
    sgo.compute = true;
    sgo.fuseChains = fuse;
    sgo.batch = static_cast<uint32_t>(batch);
    return sgo;
}

static void makeIO(snn_model* m, bool half = false) {
    m->inputs = ImageTextureArray(m->context);
    m->outputs = ImageTextureArray(m->context);
    m->inputs.push_back(ImageTextureFactory::createImageTexture(m->context, {static_cast<uint32_t>(m->inW), static_cast<uint32_t>(m->inH),
                                                                             static_cast<uint32_t>(UP_DIV(m->inC, 4)), static_cast<uint32_t>(m->batch)},
                                                                half ? ColorFormat::RGBA16F : ColorFormat::RGBA32F, nullptr, static_cast<uint32_t>(m->inC)));
    m->outputs.allocate(1);
}

extern "C" {

int snn_model_create(const char* json_path, int device, int in_w, int in_h, int in_c, int dump_outputs, int fuse_chains, int profiling,
                     snn_model** out) {
    return snn_model_create2(json_path, device, in_w, in_h, in_c, dump_outputs, fuse_chains, profiling, 0, out);
}

int snn_model_create2(const char* json_path, int device, int in_w, int in_h, int in_c, int dump_outputs, int fuse_chains, int profiling,
                      int prefer_half, snn_model** out) {
    return snn_model_create3(json_path, device, in_w, in_h, in_c, dump_outputs, fuse_chains, profiling, prefer_half, 0, out);
}

int snn_model_create3(const char* json_path, int device, int in_w, int in_h, int in_c, int dump_outputs, int fuse_chains, int profiling,
                      int prefer_half, int capture_graph, snn_model** out) {
    return snn_model_create4(json_path, device, in_w, in_h, in_c, dump_outputs, fuse_chains, profiling, prefer_half, capture_graph, 1, out);
}

int snn_model_create4(const char* json_path, int device, int in_w, int in_h, int in_c, int dump_outputs, int fuse_chains, int profiling,
                      int prefer_half, int capture_graph, int batch, snn_model** out) {
    if (!json_path || !out || batch < 1 || in_w < 1 || in_h < 1 || in_c < 1) return -1;
    const bool half = prefer_half != 0;
    auto* m = new snn_model();
    // C++ exceptions (std::bad_alloc, a parser's std::out_of_range, ...) must not unwind through the C boundary: report -2 and free what was built.
    // (SNN_RIP is the reference's abort-on-fatal-error convention, utils.h:57-62: it terminates the process and is not an exception.)
    try {
        m->batch = batch;
        m->context = createHipContext(device);
        m->inW = in_w;
        m->inH = in_h;
        m->inC = in_c;
        dp::ShaderGenOptions sgo = makeOptions(in_w, in_h, in_c, fuse_chains != 0, half, batch);
        auto layers = dp::loadFromJsonModel(json_path, false, sgo.mrtMode, sgo.weightMode, half); // preferHp: weights truncated to fp16 (Q13)
        MixedInferenceCore::CreationParameters cp;
        static_cast<InferenceGraph&>(cp) = dp::generateInferenceGraph(layers, sgo);
        cp.dumpOutputs = dump_outputs != 0;
        cp.fuseChains = fuse_chains != 0;
        cp.profiling = profiling != 0;
        cp.captureGraph = capture_graph != 0;
        m->core = MixedInferenceCore::create(m->context, cp);
        makeIO(m, half);
        m->half = half;
    } catch (const std::exception& e) {
        SNN_LOGE("snn_model_create: %s", e.what());
        snn_model_destroy(m);
        return -2;
    } catch (...) {
        snn_model_destroy(m);
        return -2;
    }
    *out = m;
    return 0;
}

int snn_model_destroy(snn_model* m) {
    if (!m) return 0;
    m->core.reset();
    m->inputs = ImageTextureArray(nullptr);
    m->outputs = ImageTextureArray(nullptr);
    delete m->context;
    delete m;
    return 0;
}

int snn_model_upload_input(snn_model* m, const float* nhwc) {
    m->inputs[0].uploadNHWC(nhwc);
    return 0;
}

static int runModel(snn_model* m, bool deferSync) {
    try {
        MixedInferenceCore::RunParameters rp;
        rp.inputImages = &m->inputs;
        rp.outputImages = &m->outputs;
        rp.modelOutput.modelType = m->modelOutput.modelType;
        rp.deferSync = deferSync;
        m->core->run(rp);
        m->modelOutput = rp.modelOutput;
    } catch (const std::exception& e) {
        SNN_LOGE("snn_model_run: %s", e.what());
        return -2;
    } catch (...) {
        return -2;
    }
    return 0;
}

int snn_model_run(snn_model* m) { return runModel(m, false); }

int snn_model_run_async(snn_model* m) { return runModel(m, true); }

int snn_model_sync(snn_model* m) { return m->core->sync() ? 0 : -1; }

int snn_json_number(const char* text, double* out) {
    if (!text || !out) return -1;
    json::Value v;
    if (!json::parse(v, text).empty() || !v.isNumber()) return -1;
    *out = v.num;
    return 0;
}

int snn_model_set_type(snn_model* m, int model_type) {
    if (model_type < 0 || model_type > 3) return -1;
    m->modelOutput.modelType = static_cast<ModelType>(model_type);
    return 0;
}

int snn_model_classifier_output(snn_model* m) { return m->modelOutput.classifierOutput; }

static int copyRows(const std::vector<std::vector<float>>& rows, float* rows6, int max_rows) {
    int n = 0;
    for (auto& r : rows) {
        if (n >= max_rows) break;
        for (size_t k = 0; k < 6 && k < r.size(); ++k) rows6[n * 6 + static_cast<int>(k)] = r[k];
        ++n;
    }
    return n;
}

int snn_model_detections(snn_model* m, float* rows6, int max_rows) { return copyRows(m->modelOutput.detectionOutput, rows6, max_rows); }

int snn_yolo_decode(const float* head_coarse, const float* head_fine, int net_size, float* rows6, int max_rows) {
    return copyRows(dp::YOLOLayer::decode({head_coarse, head_fine}, net_size), rows6, max_rows);
}

int snn_model_upload_input_u8(snn_model* m, const unsigned char* pixels, int w, int h, int channels, const float means[4], const float norms[4],
                              const float resize_means[4], const float resize_norms[4]) {
    if (m->inC != 4) return -1;
    auto arr = [](const float* p, float dflt) { return std::array<float, 4>{{p ? p[0] : dflt, p ? p[1] : dflt, p ? p[2] : dflt, p ? p[3] : dflt}}; };
    ImageTexture& t = m->inputs[0];
    t.loadU8AndNormalize(pixels, static_cast<uint32_t>(w), static_cast<uint32_t>(h), static_cast<uint32_t>(channels), arr(means, 0.0f), arr(norms, 1.0f));
    t.resize(static_cast<float>(w) / static_cast<float>(m->inW), static_cast<float>(h) / static_cast<float>(m->inH), arr(resize_means, 0.0f), arr(resize_norms, 1.0f));
    return (static_cast<int>(t.width()) == m->inW && static_cast<int>(t.height()) == m->inH) ? 0 : -2;
}

static ImageTexture& lastOutput(snn_model* m) { return m->core->stage(m->core->numStages() - 1).stageOutputs[0]; }

int snn_model_output_dims(snn_model* m, int hwc[3]) {
    ImageTexture& t = lastOutput(m);
    hwc[0] = static_cast<int>(t.height());
    hwc[1] = static_cast<int>(t.width());
    hwc[2] = static_cast<int>(t.channels());
    return 0;
}

int snn_model_batch(snn_model* m) { return m->batch; }

snnhip_ctx* snn_model_hip_ctx(snn_model* m) {
    auto* hc = dynamic_cast<HipContext*>(m->context);
    return hc ? hc->ctx : nullptr;
}

snnhip_tensor* snn_model_output_tensor(snn_model* m) { return lastOutput(m).tensor(); }

int snn_model_download_output(snn_model* m, float* nhwc) {
    lastOutput(m).downloadNHWC(nhwc);
    return 0;
}

int snn_model_num_stages(snn_model* m) { return static_cast<int>(m->core->numStages()); }

int snn_model_stage_info(snn_model* m, int stage, char* name, int name_len, int hwc[3], int* fused_away) {
    RenderStage& s = m->core->stage(static_cast<size_t>(stage));
    snprintf(name, static_cast<size_t>(name_len), "%s", s.layer->name.c_str());
    hwc[0] = static_cast<int>(s.layer->outputDesc.height);
    hwc[1] = static_cast<int>(s.layer->outputDesc.width);
    hwc[2] = static_cast<int>(s.layer->outputDesc.channels);
    *fused_away = (s.fusedAway ? 1 : 0) | (s.sideOfPrevious ? 2 : 0) | (s.groupWithPrevious ? 4 : 0);
    return 0;
}

int snn_model_download_stage(snn_model* m, int stage, float* nhwc) {
    RenderStage& s = m->core->stage(static_cast<size_t>(stage));
    if (s.layer->isInputLayer || s.stageOutputs.size() == 0 || !s.stageOutputs[0].isValid()) return -1;
    s.stageOutputs[0].downloadNHWC(nhwc);
    return 0;
}

int snn_model_describe(snn_model* m, char* buf, int buflen) {
    snprintf(buf, static_cast<size_t>(buflen), "%s", m->core->describe().c_str());
    return 0;
}

// the plan a stage really launches (nullptr: input layer, CPU stage, or folded into a later stage's fused plan)
static snnhip_plan* stagePlan(snn_model* m, int stage) {
    if (stage < 0 || stage >= static_cast<int>(m->core->numStages())) return nullptr;
    RenderStage& s = m->core->stage(static_cast<size_t>(stage));
    auto* ml = static_cast<dp::GenericModelLayer*>(s.layer->modelLayer);
    if (!ml || s.layer->isInputLayer || s.fusedAway || ml->getRenderPasses().size() != 1) return nullptr;
    auto* rp = dynamic_cast<dp::HipRenderPass*>(ml->getRenderPasses()[0].get());
    return (rp && !rp->skip) ? rp->plan : nullptr;
}

int snn_model_stage_plan_steps(snn_model* m, int stage) {
    snnhip_plan* p = stagePlan(m, stage);
    return p ? snnhip_plan_num_steps(p) : 0;
}

int snn_model_stage_plan_step(snn_model* m, int stage, int step, char* desc, int desc_len, double* flops, double* bytes) {
    snnhip_plan* p = stagePlan(m, stage);
    if (!p) return -1;
    if (desc && desc_len > 0 && snnhip_plan_step_describe(p, step, desc, static_cast<size_t>(desc_len)) != SNNHIP_OK) return -1;
    return snnhip_plan_step_cost(p, step, flops, bytes) == SNNHIP_OK ? 0 : -1;
}

int snn_model_profile_enable(snn_model* m, int enable) {
    m->core->suspendReplay(enable != 0); // a replayed hipGraph never calls the plans: profiled inferences run launch by launch
    for (int i = 0; i < static_cast<int>(m->core->numStages()); ++i)
        if (snnhip_plan* p = stagePlan(m, i))
            if (snnhip_plan_profile_enable(p, enable) != SNNHIP_OK) return -1;
    return 0;
}

int snn_model_suspend_replay(snn_model* m, int suspend) {
    m->core->suspendReplay(suspend != 0);
    return 0;
}

int snn_model_profile_read(snn_model* m, int stage, int step, double* total_ms, int* launches) {
    snnhip_plan* p = stagePlan(m, stage);
    if (!p) return -1;
    return snnhip_plan_profile_read(p, step, total_ms, launches) == SNNHIP_OK ? 0 : -1;
}

int snn_model_cost(snn_model* m, double* flops, double* bytes) {
    double f = 0, b = 0;
    for (int i = 0; i < static_cast<int>(m->core->numStages()); ++i)
        if (snnhip_plan* p = stagePlan(m, i)) {
            double pf = 0, pb = 0;
            if (snnhip_plan_cost(p, &pf, &pb) != SNNHIP_OK) return -1;
            f += pf;
            b += pb;
        }
    if (flops) *flops = f;
    if (bytes) *bytes = b;
    return 0;
}

int snn_model_time_stats(snn_model* m, char* names, int names_len, double* ms, int max_entries) {
    std::map<std::string, std::vector<double>> t;
    m->core->writeTimeStat(t);
    int n = 0;
    std::string all;
    for (auto& kv : t) {
        if (n >= max_entries) break;
        ms[n++] = kv.second.empty() ? 0.0 : kv.second.back();
        all += kv.first + "\n";
    }
    snprintf(names, static_cast<size_t>(names_len), "%s", all.c_str());
    return n;
}

int snn_conv_test_with_layer(int device, const float* input_hwc, const float* weights_oihw, const float* bias, int width, int height, int in_channels,
                             int out_channels, int kernel, int stride, int pad, int use_bn, const float* bn_gamma, const float* bn_mean,
                             const float* bn_var, const float* bn_beta, char* dump_path, int dump_path_len) {
    // ---- shaderUnitTest.cpp:192-230: hand-built InputLayer + Conv2D
    dp::InputLayerDesc inputDesc;
    inputDesc.inputHeight = static_cast<uint32_t>(width); // (sic) the reference swaps the two, shaderUnitTest.cpp:193-194
    inputDesc.inputWidth = static_cast<uint32_t>(height);
    inputDesc.inputChannels = static_cast<uint32_t>(in_channels);
    inputDesc.numInputPlanes = inputDesc.numOutputPlanes = static_cast<uint32_t>(in_channels);
    inputDesc.isInputLayer = true;
    auto inputLayer = std::make_shared<dp::InputLayerLayer>(inputDesc);
    inputLayer->setName("resnet18_cifar10_0223.json layer [00] InputLayerLayer");

    dp::Conv2DDesc desc;
    desc.isRange01 = false;
    desc.numOutputPlanes = static_cast<uint32_t>(out_channels);
    desc.numInputPlanes = static_cast<uint32_t>(in_channels);
    for (int p = 0; p < in_channels * out_channels; ++p) {
        WeightMat m(kernel, kernel);
        memcpy(m.data.data(), weights_oihw + static_cast<size_t>(p) * kernel * kernel, sizeof(float) * kernel * kernel);
        desc.weightsCvM.push_back(m);
    }
    for (int o = 0; o < out_channels; ++o) desc.biases.push_back(bias ? bias[o] : 0.0);
    desc.activation = "";
    desc.kernelSize = static_cast<uint32_t>(kernel);
    desc.stride = static_cast<uint32_t>(stride);
    desc.useBatchNormalization = use_bn != 0;
    if (use_bn) {
        desc.batchNormalization["gamma"].assign(bn_gamma, bn_gamma + out_channels);
        desc.batchNormalization["movingMean"].assign(bn_mean, bn_mean + out_channels);
        desc.batchNormalization["movingVariance"].assign(bn_var, bn_var + out_channels);
        desc.batchNormalization["beta"].assign(bn_beta, bn_beta + out_channels);
    }
    desc.useMultiInputs = false;
    desc.padding = "same";
    desc.paddingT = desc.paddingB = desc.paddingL = desc.paddingR = std::to_string(kernel / 2);
    desc.paddingMode = pad == 0 ? "constant" : pad == 1 ? "replicate" : "reflect";
    desc.weightMode = WeightAccessMethod::TEXTURES;
    std::shared_ptr<dp::GenericModelLayer> layer(dp::Conv2DCreator1(std::move(desc), false));
    layer->prevLayers.push_back(inputLayer);
    layer->setName("resnet18_cifar10_0223.json layer [01] Conv2D");
    inputLayer->nextLayers.push_back(layer);
    std::vector<std::shared_ptr<dp::GenericModelLayer>> layers{inputLayer, layer};

    // ---- :248-273: options, graph, textures, create, run
    snn_model m;
    m.context = createHipContext(device);
    m.inW = width;
    m.inH = height;
    m.inC = in_channels;
    dp::ShaderGenOptions sgo = makeOptions(width, height, in_channels, false);
    MixedInferenceCore::CreationParameters graph;
    static_cast<InferenceGraph&>(graph) = dp::generateInferenceGraph(layers, sgo);
    graph.dumpOutputs = true;
    makeIO(&m);
    m.inputs[0].uploadNHWC(input_hwc); // hwcToC4 + upload in the reference (:244-246, :265)
    m.core = MixedInferenceCore::create(m.context, graph);
    MixedInferenceCore::RunParameters rp;
    rp.inputImages = &m.inputs;
    rp.outputImages = &m.outputs;
    m.core->run(rp);
    snprintf(dump_path, static_cast<size_t>(dump_path_len), "%s/%s pass[0].dump", outputDir(), layer->getName().c_str()); // :275
    m.core.reset();
    m.inputs = ImageTextureArray(nullptr);
    m.outputs = ImageTextureArray(nullptr);
    layers.clear();
    layer.reset();
    inputLayer->nextLayers.clear();
    inputLayer.reset();
    delete m.context;
    return 0;
}

int snn_graph_summary(const char* json_path, int in_w, int in_h, int in_c, char* buf, int buflen) {
    dp::ShaderGenOptions sgo = makeOptions(in_w, in_h, in_c, true);
    auto layers = dp::loadFromJsonModel(json_path, false, sgo.mrtMode, sgo.weightMode, false);
    InferenceGraph g = dp::generateInferenceGraph(layers, sgo);
    std::string out;
    for (size_t i = 0; i < g.layers.size(); ++i) {
        auto& l = *g.layers[i];
        std::string ins;
        for (auto& r : l.inputRefs) ins += (ins.empty() ? "" : ",") + std::to_string(r.index);
        out += formatString("%zu|%s|%d|%ux%ux%u|%s\n", i, l.name.c_str(), static_cast<int>(l.layerLoc), l.outputDesc.width, l.outputDesc.height,
                            l.outputDesc.channels, ins.c_str());
    }
    snprintf(buf, static_cast<size_t>(buflen), "%s", out.c_str());
    return static_cast<int>(g.layers.size());
}

int snn_dump_read(const char* path, int whdc[4], float* out, long out_floats) {
    RawImage img = RawImage::loadFromBIN(path);
    whdc[0] = static_cast<int>(img.width());
    whdc[1] = static_cast<int>(img.height());
    whdc[2] = static_cast<int>(img.depth());
    whdc[3] = static_cast<int>(img.channels());
    if (out) {
        const long n = static_cast<long>(img.size() / sizeof(float));
        if (n > out_floats) return -1;
        memcpy(out, img.data(), img.size());
    }
    return 0;
}

} // extern "C"
