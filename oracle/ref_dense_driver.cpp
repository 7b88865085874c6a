/*
 * ref_dense_driver.cpp -- runs the REFERENCE's own CPU operator path (CPUCommonUtil<float>, Eigen) on a dense layer.
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile straight from the sources where they lie under
 * /root/reference (core/src/ic2/cpulayer.h + core/inc + the vendored core/3rdparty/eigen-3.4.0); outputs go to
 * oracle/_ref/ only and nothing from the reference is copied into this repository.
 *
 * cpulayer.h is header-only; the four functions below are declared in the reference's snn/utils.h and defined in
 * core/src/utils.cpp, a translation unit that cannot be built here (it includes cmrc-generated asset code).  They
 * are logging / abort / fp16 helpers that are not on the dense arithmetic path (fp16 conversion is only reached for
 * RGBA16F textures, which this driver never creates); the driver supplies link-time definitions for them.
 *
 * stdin : "<In> <Out> <activation|-> <leakyAlpha>\n" then Out*In kernel floats (the parser's FLAT kernel array),
 *         Out bias floats, In input floats.
 * stdout: Out result floats, %.9g, one per line.
 * Timing mode (bench.py's cpu_baseline leg for the dense rows): `ref_dense --time <In> <Out> <activation|-> <reps>`
 *         runs the same call sequence `reps` times on fixed pseudo-random data (a new CPUCommonUtil per call, as
 *         DenseLayer::computeImageTexture does) and prints "<seconds per call> <checksum>".
 * Mirrors DenseLayer::computeImageTexture (core/src/ic2/denselayer.cpp:27-38) and ModelParser::getDenseLayer's
 * [In][Out]-shaped row split of the flat kernel (core/src/ic2/modelparser.cpp:527-535).
 */
// The reference's own TUs reach cpulayer.h through genericlayer.h (-> snn/image.h, <cfloat>); reproduce that order.
#include <cfloat>
#include "snn/image.h"
#include "ic2/cpulayer.h"

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

namespace snn {
void log(const char*, int, const char*, int, int, const char* format, ...) {
    va_list args;
    va_start(args, format);
    vfprintf(stderr, format, args);
    va_end(args);
    fputc('\n', stderr);
}
bool isLoggable(int, int&, const char*) { return false; }
void rip() { abort(); }
float convertToHighPrecision(uint16_t) { abort(); }
} // namespace snn

static int time_mode(int In, int Out, std::string act, int reps) {
    if (act == "-") act = "";
    std::vector<std::vector<float>> weights(In, std::vector<float>(Out));
    std::vector<float> bias(Out), x(In);
    unsigned s = 12345u;
    auto rnd = [&s]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& row : weights)
        for (auto& v : row) v = rnd() * 0.1f;
    for (auto& v : bias) v = rnd();
    for (auto& v : x) v = rnd();
    auto transformMats = std::pair<std::vector<std::vector<float>>, std::vector<float>>(weights, bias);
    double checksum = 0.0;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
        std::vector<std::vector<float>> inputMat(1, x);
        auto cpuL = snn::dp::CPUCommonUtil<float> {act, 0.1f, true};
        cpuL.inputMat.emplace(inputMat);
        cpuL.run(transformMats);
        auto out = cpuL.getOutputs();
        checksum += out[0][r % Out];
    }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%.9g %.9g\n", sec / reps, checksum);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 6 && strcmp(argv[1], "--time") == 0) return time_mode(atoi(argv[2]), atoi(argv[3]), argv[4], atoi(argv[5]));
    int In, Out;
    std::string act;
    float alpha;
    if (!(std::cin >> In >> Out >> act >> alpha)) return 2;
    if (act == "-") act = "";
    std::vector<float> flat((size_t) In * Out), bias(Out), x(In);
    for (auto& v : flat) std::cin >> v;
    for (auto& v : bias) std::cin >> v;
    for (auto& v : x) std::cin >> v;

    // ModelParser::getDenseLayer: numInputUnits rows of numOutputUnits consecutive floats (modelparser.cpp:527-535)
    std::vector<std::vector<float>> weights(In, std::vector<float>(Out));
    size_t k = 0;
    for (int i = 0; i < In; ++i)
        for (int j = 0; j < Out; ++j) weights[i][j] = flat[k++];

    // DenseLayer::computeImageTexture (denselayer.cpp:27-38)
    std::vector<std::vector<float>> inputMat(1, x);
    auto cpuL = snn::dp::CPUCommonUtil<float> {act, alpha, true};
    auto transformMats = std::pair<std::vector<std::vector<float>>, std::vector<float>>(weights, bias);
    cpuL.inputMat.emplace(inputMat);
    cpuL.run(transformMats);
    auto out = cpuL.getOutputs();
    for (auto& row : out)
        for (float v : row) printf("%.9g\n", v);
    return 0;
}
