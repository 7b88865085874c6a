// ic2/modelparser.h -- parser of the reference's JSON static-graph format (reference core/src/ic2/modelparser.{h,cpp}).
// Same getters, same key names, same error behaviour (getters log and return -1 on malformed input); the only interface
// change is that convolution weights come back as snn::WeightMat (a k x k float matrix) instead of cv::Mat.
#pragma once
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "ic2/json.h"
#include "snn/snn.h"

namespace snn {

// Minimal stand-in for the cv::Mat(k, k, CV_32FC1) the reference stores per (oc, ic) pair (genericlayer.h:142,
// shaderUnitTest.cpp:208): rows, cols, at<float>(r, c) / at<float>(i), data.
struct WeightMat {
    int rows = 0, cols = 0;
    std::vector<float> data;
    WeightMat() = default;
    WeightMat(int r, int c) : rows(r), cols(c), data(static_cast<size_t>(r) * c, 0.0f) {}
    template <typename T>
    T& at(int r, int c) {
        return data[static_cast<size_t>(r) * cols + c];
    }
    template <typename T>
    const T& at(int r, int c) const {
        return data[static_cast<size_t>(r) * cols + c];
    }
    template <typename T>
    const T& at(int i) const {
        return data[i];
    }
};

namespace dp {

class ModelParser {
public:
    struct CreationParameters {
        const std::string filename;
        bool preferHp;
        MRTMode mrtMode;
        WeightAccessMethod weightMode;
    };
    explicit ModelParser(const CreationParameters cp);
    ~ModelParser();

    bool isInputRange01();
    int getLayerCount();
    bool getPrecision() { return preferHp; }
    MRTMode getMRTMode() { return mrtMode; }
    WeightAccessMethod getWeightMode() { return weightMode; }
    int getInputPlanes(int layerId);
    int getOutputPlanes(int layerId);
    std::string getLayerName(int layerId);
    int getNumInbound(int layerId);
    std::vector<int> getInboundLayerId(int layerId);

    int getConvolutionLayer(int& layerId, int& numOutputPlanes, int& numInputPlanes, std::string& activation, int& kernelSize, int& stride,
                            std::vector<double>& biases, std::vector<WeightMat>& weights, bool& useBatchNormalization,
                            std::map<std::string, std::vector<float>>& batchNormalization, float& leakyReluAlpha, std::string& paddingT,
                            std::string& paddingB, std::string& paddingL, std::string& paddingR, std::string& paddingMode, bool& useMultiInputs);
    int getDepthwiseConvolutionLayer(int& layerId, int& numOutputPlanes, int& numInputPlanes, std::string& activation, int& kernelSize, int& stride,
                                     std::vector<double>& biases, std::vector<WeightMat>& weights, bool& useBatchNormalization,
                                     std::map<std::string, std::vector<float>>& batchNormalization, float& leakyReluAlpha, std::string& paddingT,
                                     std::string& paddingB, std::string& paddingL, std::string& paddingR);
    int getDenseLayer(int& layerID, int& numOutputUnits, int& numInputUnits, std::string& activation, std::vector<std::vector<float>>& weights,
                      std::vector<float>& biases, float& leakyReluAlpha);
    int getInputLayer(int& layerId, uint32_t& inputWidth, uint32_t& inputHeight, uint32_t& inputChannels, uint32_t& inputIndex);
    // ---- the operators between the convolutions: same getter names / JSON keys as the reference (modelparser.cpp:304-497,987-1201)
    int getMaxPoolLayer(int& layerID, int& numOutputPlanes, int& numInputPlanes, int& poolSize, int& stride, std::string& paddingMode,
                        std::string& paddingValue, std::string& paddingT, std::string& paddingB, std::string& paddingL, std::string& paddingR);
    int getAvgPoolLayer(int& layerID, int& numOutputPlanes, int& numInputPlanes, int& poolSize, int& stride, std::string& padding);
    int getAdaptiveAvgPoolLayer(int& layerID, int& numOutputPlanes, int& numInputPlanes, int& poolSize);
    int getAddLayer(int& layerID, std::string& activation, float& leakyReluAlpha);
    int getActivationLayer(int& layerID, std::string& activation, float& leakyReluAlpha);
    int getFlattenLayer(int& layerId, int& numOutputPlanes, int& numInputPlanes, std::string& activation);
    int getBatchNormLayer(int& layerId, int& numOutputPlanes, int& numInputPlanes, std::map<std::string, std::vector<float>>& batchNormalization,
                          std::string& activation, float& leakyReluAlpha);
    int getPaddingLayer(int& layerId, int& numOutputPlanes, int& numInputPlanes, std::string& paddingT, std::string& paddingB, std::string& paddingL,
                        std::string& paddingR, std::string& mode, float& constant);
    int getInstanceNormalizationLayer(int& layerId, int& numOutputPlanes, int& numInputPlanes, float& epsilon,
                                      std::map<std::string, std::vector<float>>& batchNormalization, std::string& activation, float& leakyReluAlpha);
    float getUpSamplingScale(int layerId);
    std::string getUpSampling2DInterpolation(int layerId);

private:
    json::Value _modelOb;
    bool preferHp;
    bool isBinWeight = false;
    std::ifstream binFile;
    MRTMode mrtMode;
    WeightAccessMethod weightMode;
    const json::Value& layer(int id) const;
    void parsePadding(const json::Value& layerObj, std::string& t, std::string& b, std::string& l, std::string& r, std::string* mode);
    void parseBatchNorm(const json::Value& layerObj, int n, bool truncate, std::map<std::string, std::vector<float>>& out);
    float readBin();
};

} // namespace dp
} // namespace snn
