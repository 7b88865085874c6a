// snn_run -- model-runner CLI over libsnn_core.so (include/snn_c.h): the role of the reference's inferenceProcessorTest
// (demo/test/unittest/inferenceProcessorTest.cpp:92-104: model, --use_half, --dump_outputs, --inner_loops) with the statistics rule of its
// benchmark loop (demo/common/inferenceProcessor.cpp:84-86,143-199: first 5 runs dropped, per-layer mean and POPULATION standard deviation).
//
//   snn_run model.json --w W --h H --c C [--batch N] [--loops L] [--half] [--dump_outputs] [--device D] [--devices D0,D1,...] [--seed S] [--no_graph]
//
// Input: synthetic U(0,1) NHWC image(s) from a fixed-seed generator (the reference feeds image files through OpenCV, which this tree does not have).
// --devices runs the batch split over several replicas through snn_pool (one host thread + context per device).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "snn_c.h"

static void usage() {
    fprintf(stderr, "usage: snn_run model.json --w W --h H --c C [--batch N] [--loops L] [--half] [--dump_outputs] [--device D] [--devices D0,D1,..] [--seed S] [--no_graph]\n");
}

int main(int argc, char** argv) {
    std::string model;
    int w = 0, h = 0, c = 0, batch = 1, loops = 1, device = 0, half = 0, dump = 0, graph = 1;
    unsigned long long seed = 7767517ull; // the reference's SRAND seed (convolutionTest.cpp:417)
    std::vector<int> devices;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* {
            if (i + 1 >= argc) {
                usage();
                exit(2);
            }
            return argv[++i];
        };
        if (a == "--w") w = atoi(next());
        else if (a == "--h") h = atoi(next());
        else if (a == "--c") c = atoi(next());
        else if (a == "--batch") batch = atoi(next());
        else if (a == "--loops" || a == "--inner_loops") loops = atoi(next());
        else if (a == "--device") device = atoi(next());
        else if (a == "--seed") seed = strtoull(next(), nullptr, 10);
        else if (a == "--half" || a == "--use_half") half = 1;
        else if (a == "--dump_outputs") dump = 1;
        else if (a == "--no_graph") graph = 0;
        else if (a == "--devices") {
            const std::string list = next();
            for (size_t at = 0; at < list.size();) {
                devices.push_back(atoi(list.c_str() + at));
                const size_t comma = list.find(',', at);
                at = comma == std::string::npos ? list.size() : comma + 1;
            }
        } else if (!a.empty() && a[0] != '-' && model.empty()) model = a;
        else {
            usage();
            return 2;
        }
    }
    if (model.empty() || w < 1 || h < 1 || c < 1 || batch < 1 || loops < 1) {
        usage();
        return 2;
    }
    // synthetic input: splitmix64 -> U(0,1)
    std::vector<float> x(static_cast<size_t>(batch) * h * w * c);
    unsigned long long s = seed;
    for (float& v : x) {
        s += 0x9E3779B97F4A7C15ull;
        unsigned long long z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        v = static_cast<float>(z >> 40) * (1.0f / 16777216.0f);
    }
    if (!devices.empty()) {
        snn_pool* pool = nullptr;
        if (snn_pool_create(model.c_str(), devices.data(), static_cast<int>(devices.size()), w, h, c, half, graph, batch, 0, &pool) != 0) {
            fprintf(stderr, "snn_run: snn_pool_create failed\n");
            return 1;
        }
        snn_pool_upload_input(pool, x.data());
        double sec = 0;
        snn_pool_run(pool, 5, &sec); // the reference's five excluded runs
        if (snn_pool_run(pool, loops, &sec) != 0) return 1;
        int hwc[3];
        snn_pool_output_dims(pool, hwc);
        std::vector<float> y(static_cast<size_t>(batch) * hwc[0] * hwc[1] * hwc[2]);
        snn_pool_download_output(pool, y.data());
        double sum = 0;
        for (float v : y) sum += v;
        printf("replicas %d | %d image(s) x %d loop(s) in %.3f ms = %.1f images/s | output %dx%dx%dx%d checksum %.6f\n", snn_pool_replicas(pool), batch, loops, 1e3 * sec,
               batch * loops / sec, batch, hwc[0], hwc[1], hwc[2], sum);
        snn_pool_destroy(pool);
        return 0;
    }
    snn_model* m = nullptr;
    // per-stage device timers need the host between stages: a timed model runs launch by launch (profiling = 1), exactly as the reference's does
    if (snn_model_create4(model.c_str(), device, w, h, c, dump, dump ? 0 : 1, 1, half, 0, batch, &m) != 0) {
        fprintf(stderr, "snn_run: cannot create the model from %s\n", model.c_str());
        return 1;
    }
    snn_model_upload_input(m, x.data());
    std::map<std::string, std::vector<double>> rows;
    std::vector<std::string> order;
    const int kDrop = 5; // NUM_EXCLUDE_FIRST_LOOPS
    for (int l = 0; l < loops + kDrop; ++l) {
        if (snn_model_run(m) != 0) return 1;
        if (l < kDrop) continue;
        char names[1 << 15];
        double ms[1024];
        const int n = snn_model_time_stats(m, names, sizeof(names), ms, 1024);
        const char* p = names; // newline-separated, in the order of ms[]
        for (int k = 0; k < n; ++k) {
            const char* e = strchr(p, '\n');
            const std::string name = e ? std::string(p, static_cast<size_t>(e - p)) : std::string(p);
            p = e ? e + 1 : p + name.size();
            if (!rows.count(name)) order.push_back(name);
            rows[name].push_back(ms[k]);
        }
    }
    printf("%-6s| %-58s| %-12s| %-12s\n", "id", "layer", "mean ms", "std dev ms");
    int id = -1;
    for (const std::string& name : order) {
        const std::vector<double>& v = rows[name];
        double mean = 0, acc = 0;
        for (double t : v) mean += t;
        mean /= static_cast<double>(v.size());
        for (double t : v) acc += (t - mean) * (t - mean);
        printf("%-6d| %-58.58s| %-12.6f| %-12.6f\n", id++, name.c_str(), mean, std::sqrt(acc / static_cast<double>(v.size())));
    }
    int hwc[3];
    snn_model_output_dims(m, hwc);
    std::vector<float> y(static_cast<size_t>(batch) * hwc[0] * hwc[1] * hwc[2]);
    snn_model_download_output(m, y.data());
    double sum = 0;
    for (float v : y) sum += v;
    printf("output %dx%dx%dx%d checksum %.6f (%d timed loop(s), first %d dropped)%s\n", batch, hwc[0], hwc[1], hwc[2], sum, loops, kDrop,
           dump ? "; layer dumps written to $SNN_OUTPUT_DIR" : "");
    snn_model_destroy(m);
    return 0;
}
