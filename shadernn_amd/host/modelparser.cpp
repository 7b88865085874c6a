// modelparser.cpp -- reads the reference's JSON model format (reference core/src/ic2/modelparser.cpp).
// Key names, defaults and error behaviour (log + return -1) follow the reference getter by getter; citations inline.
// Deviation: a "bin_file_name" side file is resolved next to the JSON file (the reference builds a path from its
// MODEL_DIR macro and the current directory, modelparser.cpp:234-257).
#include <sstream>

#include "ic2/modelparser.h"

using namespace snn;
using namespace snn::dp;

static std::string dirOf(const std::string& path) {
    const size_t p = path.find_last_of('/');
    return p == std::string::npos ? std::string(".") : path.substr(0, p);
}

ModelParser::ModelParser(const CreationParameters cp) : preferHp(cp.preferHp), mrtMode(cp.mrtMode), weightMode(cp.weightMode) {
    std::ifstream f(cp.filename, std::ios::binary);
    if (!f.good()) SNN_RIP("ModelParser:: Could not load JSON file %s", cp.filename.c_str()); // :226-228
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string err = json::parse(_modelOb, ss.str());
    if (!err.empty()) SNN_RIP("ModelParser:: Could not parse JSON file %s (%s)", cp.filename.c_str(), err.c_str()); // :231-233
    const json::Value& numNode = _modelOb.at("numLayers");
    if (numNode.has("bin_file_name")) { // :236-257
        const std::string path = dirOf(cp.filename) + "/" + numNode.at("bin_file_name").asString();
        isBinWeight = true;
        binFile.open(path, std::ios::binary);
        if (!binFile.good()) SNN_RIP("open %s failed", path.c_str());
    }
}

ModelParser::~ModelParser() {
    if (isBinWeight) binFile.close();
}

const json::Value& ModelParser::layer(int id) const { return _modelOb.at("Layer_" + std::to_string(id)); }

float ModelParser::readBin() {
    float v = 0.0f;
    binFile.read(reinterpret_cast<char*>(&v), sizeof(float));
    if (!binFile.good()) throw std::runtime_error("weight .bin file is too short");
    return v;
}

bool ModelParser::isInputRange01() { // :31-35
    return _modelOb.has("inputRange") && _modelOb.at("inputRange").isString() && _modelOb.at("inputRange").asString() == "[0,1]";
}

int ModelParser::getLayerCount() { return static_cast<int>(_modelOb.at("numLayers").at("count").asNumber()); } // :39-44

int ModelParser::getNumInbound(int layerId) { return static_cast<int>(layer(layerId).at("numInputs").asNumber()); } // :126-131

int ModelParser::getInputPlanes(int layerId) { // :61-69
    if (getNumInbound(layerId) != 0) return static_cast<int>(layer(layerId).at("inputPlanes").asNumber());
    return 0;
}

int ModelParser::getOutputPlanes(int layerId) { return static_cast<int>(layer(layerId).at("outputPlanes").asNumber()); } // :71-76

std::string ModelParser::getLayerName(int layerId) { // :78-86: Keras Lambda layers are dispatched by NAME
    std::string cls = layer(layerId).at("type").asString();
    if (cls == "Lambda") cls = layer(layerId).at("name").asString();
    return cls;
}

std::vector<int> ModelParser::getInboundLayerId(int layerId) { // :133-143
    const int n = getNumInbound(layerId);
    std::vector<int> ids;
    const json::Array& nodes = layer(layerId).at("inputId").asArray();
    for (int i = 0; i < n; ++i) ids.push_back(static_cast<int>(nodes.at(static_cast<size_t>(i)).asNumber()));
    return ids;
}

int ModelParser::getInputLayer(int& layerId, uint32_t& inputWidth, uint32_t& inputHeight, uint32_t& inputChannels, uint32_t& inputIndex) { // :480-497
    try {
        const json::Value& o = layer(layerId);
        inputWidth = static_cast<uint32_t>(o.at("Input Width").asNumber());
        inputHeight = static_cast<uint32_t>(o.at("Input Height").asNumber());
        inputChannels = static_cast<uint32_t>(o.at("outputPlanes").asNumber());
        if (o.has("inputIndex")) inputIndex = static_cast<uint32_t>(o.at("inputIndex").asNumber());
    } catch (std::exception& e) {
        SNN_LOGE("ModelParser::getInputLayer : Issues parsing layer %d, %s", layerId, e.what());
        inputWidth = inputHeight = inputChannels = inputIndex = 0;
    }
    return 0;
}

// "padding": "same"|"valid"|"none" | number | [t, l] | [[t, b], [l, r]] (+ "mode")   (:583-606, :793-815)
void ModelParser::parsePadding(const json::Value& o, std::string& t, std::string& b, std::string& l, std::string& r, std::string* mode) {
    const json::Value& p = o.at("padding");
    if (p.isArray()) {
        const json::Array& a = p.asArray();
        if (a.size() >= 2 && a[0].isArray()) {
            t = std::to_string(static_cast<uint32_t>(a[0].asArray().at(0).asNumber()));
            b = std::to_string(static_cast<uint32_t>(a[0].asArray().at(1).asNumber()));
            l = std::to_string(static_cast<uint32_t>(a[1].asArray().at(0).asNumber()));
            r = std::to_string(static_cast<uint32_t>(a[1].asArray().at(1).asNumber()));
            if (mode) *mode = o.at("mode").asString();
        } else {
            t = std::to_string(static_cast<uint32_t>(a.at(0).asNumber()));
            l = std::to_string(static_cast<uint32_t>(a.at(1).asNumber()));
            b = t;
            r = l;
        }
    } else {
        t = p.isNumber() ? std::to_string(static_cast<uint32_t>(p.asNumber())) : p.asString();
        b = l = r = t;
    }
}

// BN block (:685-757): JSON keys beta/gamma/moving_mean|movingMean/moving_variance|movingVariance; .bin order gamma, beta, mean, variance
void ModelParser::parseBatchNorm(const json::Value& o, int n, bool truncate, std::map<std::string, std::vector<float>>& out) {
    std::vector<float> beta(n), gamma(n), mean(n), var(n);
    auto cvt = [&](float v) { return truncate ? convertToMediumPrecision(v) : v; };
    if (isBinWeight) {
        for (int i = 0; i < n; ++i) gamma[i] = cvt(readBin());
        for (int i = 0; i < n; ++i) beta[i] = cvt(readBin());
        for (int i = 0; i < n; ++i) mean[i] = cvt(readBin());
        for (int i = 0; i < n; ++i) var[i] = cvt(readBin());
    } else {
        const json::Value& bn = o.at("batchNormalization");
        const json::Array& b = bn.at("beta").asArray();
        const json::Array& g = bn.at("gamma").asArray();
        const json::Array& m = bn.has("moving_mean") ? bn.at("moving_mean").asArray() : bn.at("movingMean").asArray();
        const json::Array& v = bn.has("moving_variance") ? bn.at("moving_variance").asArray() : bn.at("movingVariance").asArray();
        for (int i = 0; i < n; ++i) {
            beta[i] = cvt(static_cast<float>(b.at(i).asNumber()));
            gamma[i] = cvt(static_cast<float>(g.at(i).asNumber()));
            mean[i] = cvt(static_cast<float>(m.at(i).asNumber()));
            var[i] = cvt(static_cast<float>(v.at(i).asNumber()));
        }
    }
    out["beta"] = beta;
    out["gamma"] = gamma;
    out["movingMean"] = mean;
    out["movingVariance"] = var;
}

int ModelParser::getConvolutionLayer(int& layerId, int& numOutputPlanes, int& numInputPlanes, std::string& activation, int& kernelSize, int& stride,
                                     std::vector<double>& biases, std::vector<WeightMat>& weights, bool& useBatchNormalization,
                                     std::map<std::string, std::vector<float>>& batchNormalization, float& leakyReluAlpha, std::string& paddingT,
                                     std::string& paddingB, std::string& paddingL, std::string& paddingR, std::string& paddingMode,
                                     bool& useMultiInputs) { // :574-781
    try {
        const json::Value& o = layer(layerId);
        numOutputPlanes = static_cast<int>(o.at("outputPlanes").asNumber());
        numInputPlanes = static_cast<int>(o.at("inputPlanes").asNumber());
        activation = o.at("activation").asString();
        parsePadding(o, paddingT, paddingB, paddingL, paddingR, &paddingMode);
        kernelSize = static_cast<int>(o.at("kernel_size").asNumber());
        stride = static_cast<int>(o.at("strides").asNumber());
        useMultiInputs = o.has("use_multi_inputs") && o.at("use_multi_inputs").asString() == "True";

        // kernel: flat OIHW -> OC*IC matrices of k x k (:617-658); fp16 mode truncates every weight (:628-630)
        weights.assign(static_cast<size_t>(numInputPlanes) * numOutputPlanes, WeightMat(kernelSize, kernelSize));
        const json::Array* arr = isBinWeight ? nullptr : &o.at("weights").at("kernel").asArray();
        size_t e = 0;
        for (int i = 0; i < numOutputPlanes; ++i)
            for (int j = 0; j < numInputPlanes; ++j) {
                WeightMat& m = weights[static_cast<size_t>(i) * numInputPlanes + j];
                for (int r = 0; r < kernelSize; ++r)
                    for (int c = 0; c < kernelSize; ++c) {
                        float v = isBinWeight ? readBin() : static_cast<float>(arr->at(e).asNumber());
                        ++e;
                        if (preferHp) v = convertToMediumPrecision(v);
                        m.at<float>(r, c) = v;
                    }
            }
        biases.assign(static_cast<size_t>(numOutputPlanes), 0.0); // always sized => the conv layer always reports useBias (:660)
        if (o.at("useBias").asString() == "True") {
            const json::Array* b = isBinWeight ? nullptr : &o.at("weights").at("bias").asArray();
            for (int i = 0; i < numOutputPlanes; ++i) {
                float v = isBinWeight ? readBin() : static_cast<float>(b->at(static_cast<size_t>(i)).asNumber());
                biases[static_cast<size_t>(i)] = preferHp ? convertToMediumPrecision(v) : (isBinWeight ? v : b->at(static_cast<size_t>(i)).asNumber());
            }
        }
        useBatchNormalization = o.at("useBatchNormalization").asString() == "True";
        if (useBatchNormalization) parseBatchNorm(o, numOutputPlanes, preferHp, batchNormalization);
        if (activation == "leakyRelu") { // :764-771
            leakyReluAlpha = static_cast<float>(o.has("leakyReluAlpha") ? o.at("leakyReluAlpha").asNumber() : o.at("alpha").asNumber());
            if (preferHp) leakyReluAlpha = convertToMediumPrecision(leakyReluAlpha);
        }
    } catch (std::exception& e) {
        SNN_LOGE("ModelParser::getConvolutionLayer : Issues parsing layer %d, %s", layerId, e.what());
        return -1;
    }
    return 0;
}

int ModelParser::getDepthwiseConvolutionLayer(int& layerId, int& numOutputPlanes, int& numInputPlanes, std::string& activation, int& kernelSize,
                                              int& stride, std::vector<double>& biases, std::vector<WeightMat>& weights, bool& useBatchNormalization,
                                              std::map<std::string, std::vector<float>>& batchNormalization, float& leakyReluAlpha,
                                              std::string& paddingT, std::string& paddingB, std::string& paddingL, std::string& paddingR) { // :783-985
    try {
        const json::Value& o = layer(layerId);
        numOutputPlanes = static_cast<int>(o.at("outputPlanes").asNumber());
        numInputPlanes = static_cast<int>(o.at("inputPlanes").asNumber());
        activation = o.at("activation").asString();
        parsePadding(o, paddingT, paddingB, paddingL, paddingR, nullptr);
        kernelSize = static_cast<int>(o.at("kernel_size").asNumber());
        stride = static_cast<int>(o.at("strides").asNumber());
        weights.assign(static_cast<size_t>(numInputPlanes), WeightMat(kernelSize, kernelSize));
        const int plane = kernelSize * kernelSize;
        if (isBinWeight) { // .bin kernel is CHW (:826-840)
            for (int c = 0; c < numInputPlanes; ++c)
                for (int i = 0; i < plane; ++i) {
                    float v = readBin();
                    weights[static_cast<size_t>(c)].data[static_cast<size_t>(i)] = preferHp ? convertToMediumPrecision(v) : v;
                }
        } else { // JSON kernel is flat HWC (:842-850)
            const json::Array& arr = o.at("weights").at("kernel").asArray();
            for (int i = 0; i < plane; ++i)
                for (int c = 0; c < numInputPlanes; ++c) {
                    float v = static_cast<float>(arr.at(static_cast<size_t>(i) * numInputPlanes + c).asNumber());
                    weights[static_cast<size_t>(c)].data[static_cast<size_t>(i)] = preferHp ? convertToMediumPrecision(v) : v;
                }
        }
        biases.assign(static_cast<size_t>(numOutputPlanes), 0.0);
        if (o.at("useBias").asString() == "True") {
            const json::Array* b = isBinWeight ? nullptr : &o.at("weights").at("bias").asArray();
            for (int i = 0; i < numOutputPlanes; ++i) {
                float v = isBinWeight ? readBin() : static_cast<float>(b->at(static_cast<size_t>(i)).asNumber());
                biases[static_cast<size_t>(i)] = preferHp ? convertToMediumPrecision(v) : v;
            }
        }
        useBatchNormalization = o.at("useBatchNormalization").asString() == "True";
        if (useBatchNormalization) parseBatchNorm(o, numOutputPlanes, preferHp && !isBinWeight, batchNormalization); // .bin BN is not truncated (:894-920)
        if (activation == "leakyRelu") {
            leakyReluAlpha = static_cast<float>(o.has("leakyReluAlpha") ? o.at("leakyReluAlpha").asNumber() : o.at("alpha").asNumber());
            if (preferHp) leakyReluAlpha = convertToMediumPrecision(leakyReluAlpha);
        }
    } catch (std::exception& e) {
        SNN_LOGE("ModelParser::getConvolutionLayer : Issues parsing layer %d, %s", layerId, e.what());
        return -1;
    }
    return 0;
}

int ModelParser::getDenseLayer(int& layerID, int& numOutputUnits, int& numInputUnits, std::string& activation, std::vector<std::vector<float>>& weights,
                               std::vector<float>& biases, float& leakyReluAlpha) { // :499-572
    try {
        const json::Value& o = layer(layerID);
        const int numOutputPlanes = static_cast<int>(o.at("outputPlanes").asNumber());
        const int numInputPlanes = static_cast<int>(o.at("inputPlanes").asNumber());
        numOutputUnits = o.has("units") ? static_cast<int>(o.at("units").asNumber()) : numOutputPlanes;
        std::vector<std::vector<float>> mat;
        if (isBinWeight) {
            numInputUnits = numInputPlanes;
            for (int i = 0; i < numInputPlanes; ++i) {
                std::vector<float> row(static_cast<size_t>(numOutputUnits));
                for (auto& v : row) v = readBin();
                mat.push_back(row);
            }
        } else {
            const json::Array& arr = o.at("weights").at("kernel").asArray();
            numInputUnits = static_cast<int>(arr.size()) / numOutputUnits;
            size_t e = 0;
            for (int i = 0; i < numInputUnits; ++i) { // rows of numOutputUnits consecutive floats (:527-535)
                std::vector<float> row(static_cast<size_t>(numOutputUnits));
                for (auto& v : row) v = static_cast<float>(arr.at(e++).asNumber());
                mat.push_back(row);
            }
        }
        weights = std::move(mat);
        biases.clear();
        if (o.at("useBias").asString() == "True") {
            const json::Array* b = isBinWeight ? nullptr : &o.at("weights").at("bias").asArray();
            for (int i = 0; i < numOutputUnits; ++i) biases.push_back(isBinWeight ? readBin() : static_cast<float>(b->at(static_cast<size_t>(i)).asNumber()));
        } else {
            biases.assign(static_cast<size_t>(numOutputPlanes), 0.0f);
        }
        activation = o.at("activation").asString();
        if (activation == "leaky_relu") { // (sic) the dense parser looks for a different spelling than the conv parser (:558)
            try {
                leakyReluAlpha = static_cast<float>(o.has("leakyReluAlpha") ? o.at("leakyReluAlpha").asNumber() : o.at("alpha").asNumber());
            } catch (std::exception&) { leakyReluAlpha = 0.3f; }
        }
        return 0;
    } catch (std::exception& e) {
        SNN_LOGE("ModelParser::getDenseLayer : Issues parsing layer %d, %s", layerID, e.what());
        return -1;
    }
}

// ------------------------------------------------------------------------------------------------ operators between the convolutions

static int strideOf(const json::Value& o, int fallback) { // modelparser.cpp:313-334: "stride" | "strides", number or [n, n]
    for (const char* key : {"stride", "strides"}) {
        if (!o.has(key)) continue;
        const json::Value& v = o.at(key);
        if (v.isNumber()) return static_cast<int>(v.asNumber());
        if (v.isArray()) return static_cast<int>(v.asArray().at(0).asNumber());
    }
    return fallback;
}

int ModelParser::getMaxPoolLayer(int& layerID, int& numOutputPlanes, int& numInputPlanes, int& poolSize, int& stride, std::string& paddingMode,
                                 std::string& paddingValue, std::string& paddingT, std::string& paddingB, std::string& paddingL, std::string& paddingR) { // :304-371
    try {
        const json::Value& o = layer(layerID);
        numOutputPlanes = static_cast<int>(o.at("outputPlanes").asNumber());
        numInputPlanes = static_cast<int>(o.at("inputPlanes").asNumber());
        poolSize = static_cast<int>(o.at("pool").asArray().at(0).asNumber());
        stride = strideOf(o, poolSize);
        parsePadding(o, paddingT, paddingB, paddingL, paddingR, nullptr);
        const json::Value& p = o.at("padding");
        if (!p.isArray()) paddingMode = paddingT;
        paddingValue = o.has("padding_value") ? o.at("padding_value").asString() : "constant";
    } catch (std::exception& e) {
        SNN_LOGE("ModelParser::getMaxPoolLayer : Issues parsing layer %d, %s", layerID, e.what());
        return -1;
    }
    return 0;
}

int ModelParser::getAvgPoolLayer(int& layerID, int& numOutputPlanes, int& numInputPlanes, int& poolSize, int& stride, std::string& padding) { // :373-397
    try {
        const json::Value& o = layer(layerID);
        numOutputPlanes = static_cast<int>(o.at("outputPlanes").asNumber());
        numInputPlanes = static_cast<int>(o.at("inputPlanes").asNumber());
        const json::Value& pool = o.has("pool") ? o.at("pool") : o.at("pool_size");
        poolSize = static_cast<int>(pool.asArray().at(0).asNumber());
        stride = o.has("stride") ? strideOf(o, poolSize) : poolSize;
        padding = o.at("padding").asString();
    } catch (std::exception& e) {
        SNN_LOGE("ModelParser::getAvgPoolLayer : Issues parsing layer %d, %s", layerID, e.what());
        return -1;
    }
    return 0;
}

int ModelParser::getAdaptiveAvgPoolLayer(int& layerID, int& numOutputPlanes, int& numInputPlanes, int& poolSize) { // :439-451
    try {
        const json::Value& o = layer(layerID);
        numOutputPlanes = static_cast<int>(o.at("outputPlanes").asNumber());
        numInputPlanes = static_cast<int>(o.at("inputPlanes").asNumber());
        poolSize = static_cast<int>(o.at("pool").asArray().at(0).asNumber());
    } catch (std::exception& e) {
        SNN_LOGE("ModelParser::getAdaptiveAvgPoolLayer : Issues parsing layer %d, %s", layerID, e.what());
        return -1;
    }
    return 0;
}

static void activationOf(const json::Value& o, std::string& activation, float& leakyReluAlpha) { // :399-437 (note the "leaky_relu" spelling there)
    activation = o.has("activation") ? o.at("activation").asString() : "linear";
    if (activation == "leaky_relu" || activation == "leakyRelu") {
        if (o.has("leakyReluAlpha")) leakyReluAlpha = static_cast<float>(o.at("leakyReluAlpha").asNumber());
        else if (o.has("alpha")) leakyReluAlpha = static_cast<float>(o.at("alpha").asNumber());
        else leakyReluAlpha = 0.3f;
    }
}

int ModelParser::getAddLayer(int& layerID, std::string& activation, float& leakyReluAlpha) {
    try {
        activationOf(layer(layerID), activation, leakyReluAlpha);
    } catch (std::exception& e) {
        SNN_LOGE("ModelParser::getAddLayer : Issues parsing layer %d, %s", layerID, e.what());
        return -1;
    }
    return 0;
}

int ModelParser::getActivationLayer(int& layerID, std::string& activation, float& leakyReluAlpha) { return getAddLayer(layerID, activation, leakyReluAlpha); }

int ModelParser::getFlattenLayer(int& layerId, int& numOutputPlanes, int& numInputPlanes, std::string& activation) { // :453-466
    try {
        const json::Value& o = layer(layerId);
        numOutputPlanes = static_cast<int>(o.at("outputPlanes").asNumber());
        numInputPlanes = static_cast<int>(o.at("inputPlanes").asNumber());
        activation = o.has("activation") ? o.at("activation").asString() : "linear";
    } catch (std::exception& e) {
        SNN_LOGE("ModelParser::getFlattenLayer : Issues parsing layer %d, %s", layerId, e.what());
        return -1;
    }
    return 0;
}

int ModelParser::getBatchNormLayer(int& layerId, int& numOutputPlanes, int& numInputPlanes, std::map<std::string, std::vector<float>>& batchNormalization,
                                   std::string& activation, float& leakyReluAlpha) { // :1011-1110 (inline arrays; missing beta/gamma default to 0/1)
    try {
        const json::Value& o = layer(layerId);
        numOutputPlanes = static_cast<int>(o.at("outputPlanes").asNumber());
        numInputPlanes = static_cast<int>(o.at("inputPlanes").asNumber());
        const json::Value& bn = o.at("batchNormalization");
        auto cvt = [&](double v) { return preferHp ? convertToMediumPrecision(static_cast<float>(v)) : static_cast<float>(v); };
        auto arr = [&](const char* a, const char* b, bool required, float dflt) {
            std::vector<float> out(static_cast<size_t>(numOutputPlanes), cvt(dflt));
            const json::Value* v = bn.has(a) ? &bn.at(a) : (b && bn.has(b) ? &bn.at(b) : nullptr);
            if (!v) {
                if (required) throw std::runtime_error(std::string("json: missing key ") + a);
                return out;
            }
            for (int i = 0; i < numOutputPlanes; ++i) out[static_cast<size_t>(i)] = cvt(v->asArray().at(static_cast<size_t>(i)).asNumber());
            return out;
        };
        batchNormalization["beta"] = arr("beta", nullptr, false, 0.0f);
        batchNormalization["gamma"] = arr("gamma", nullptr, false, 1.0f);
        batchNormalization["movingMean"] = arr("moving_mean", "movingMean", true, 0.0f);
        batchNormalization["movingVariance"] = arr("moving_variance", "movingVariance", true, 0.0f);
        if (o.has("activation")) activation = o.at("activation").asString();
        if (activation == "leakyRelu") leakyReluAlpha = cvt(o.has("leakyReluAlpha") ? o.at("leakyReluAlpha").asNumber() : o.at("alpha").asNumber());
    } catch (std::exception& e) {
        SNN_LOGE("ModelParser::BatchNormLayer : Issues parsing layer %d, %s", layerId, e.what());
        return -1;
    }
    return 0;
}

int ModelParser::getPaddingLayer(int& layerId, int& numOutputPlanes, int& numInputPlanes, std::string& paddingT, std::string& paddingB, std::string& paddingL,
                                 std::string& paddingR, std::string& mode, float& constant) { // :1112-1147
    try {
        const json::Value& o = layer(layerId);
        numOutputPlanes = static_cast<int>(o.at("outputPlanes").asNumber());
        numInputPlanes = static_cast<int>(o.at("inputPlanes").asNumber());
        if (o.has("pads")) { // ONNX order [n, c, h, w] begin then end
            const json::Array& a = o.at("pads").asArray();
            paddingT = std::to_string(static_cast<uint32_t>(a.at(2).asNumber()));
            paddingB = std::to_string(static_cast<uint32_t>(a.at(6).asNumber()));
            paddingL = std::to_string(static_cast<uint32_t>(a.at(3).asNumber()));
            paddingR = std::to_string(static_cast<uint32_t>(a.at(7).asNumber()));
        } else {
            parsePadding(o, paddingT, paddingB, paddingL, paddingR, nullptr);
        }
        // the reference getter drops "mode"/"constant" on the floor (unnamed parameters, :1113), which leaves PadDesc::mode at its
        // "constant" default for every JSON model; here the key is honoured when present
        if (o.has("mode")) mode = o.at("mode").asString();
        if (o.has("constant")) constant = static_cast<float>(o.at("constant").asNumber());
    } catch (std::exception& e) {
        SNN_LOGE("ModelParser::getPaddingLayer : Issues parsing layer %d, %s", layerId, e.what());
        return -1;
    }
    return 0;
}

int ModelParser::getInstanceNormalizationLayer(int& layerId, int& numOutputPlanes, int& numInputPlanes, float& epsilon,
                                               std::map<std::string, std::vector<float>>& batchNormalization, std::string& activation,
                                               float& leakyReluAlpha) { // :1149-1201
    try {
        const json::Value& o = layer(layerId);
        numOutputPlanes = static_cast<int>(o.at("outputPlanes").asNumber());
        numInputPlanes = static_cast<int>(o.at("inputPlanes").asNumber());
        if (o.has("activation")) activation = o.at("activation").asString();
        epsilon = static_cast<float>(o.at("epsilon").asNumber());
        const json::Value& w = o.at("weights");
        auto cvt = [&](double v) { return preferHp ? convertToMediumPrecision(static_cast<float>(v)) : static_cast<float>(v); };
        std::vector<float> bias(static_cast<size_t>(numOutputPlanes)), scale(static_cast<size_t>(numOutputPlanes));
        for (int i = 0; i < numOutputPlanes; ++i) {
            bias[static_cast<size_t>(i)] = cvt(w.at("bias").asArray().at(static_cast<size_t>(i)).asNumber());
            scale[static_cast<size_t>(i)] = cvt(w.at("scale").asArray().at(static_cast<size_t>(i)).asNumber());
        }
        batchNormalization["beta"] = bias;
        batchNormalization["gamma"] = scale;
        if (activation == "leakyRelu") leakyReluAlpha = cvt(o.at("leakyReluAlpha").asNumber());
    } catch (std::exception& e) {
        SNN_LOGE("ModelParser::getInstanceNormLayer : Issues parsing layer %d, %s", layerId, e.what());
        return -1;
    }
    return 0;
}

float ModelParser::getUpSamplingScale(int layerId) { // :987-997
    if (getLayerName(layerId) == "UpSampling2D") return static_cast<float>(layer(layerId).at("scaleFactor").asNumber());
    SNN_LOGW("ModelParser:: accessing scale in a non upsampling2D layer");
    return 0;
}

std::string ModelParser::getUpSampling2DInterpolation(int layerId) { // :999-1009
    if (getLayerName(layerId) == "UpSampling2D") return layer(layerId).at("interpolation").asString();
    SNN_LOGW("ModelParser:: accessing interpolation in a non upsampling2D layer");
    return "";
}
