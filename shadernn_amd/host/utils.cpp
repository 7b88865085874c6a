// utils.cpp -- logging / rip / helpers (reference core/src/utils.cpp subset).
#include <cstring>

#include "snn/snn.h"
#include "snn/utils.h"

namespace snn {

static int logLevel() {
    static int lvl = [] {
        const char* e = getenv("SNN_LOG_LEVEL"); // same env var as the reference (utils.cpp:253-255)
        if (!e) return static_cast<int>(LogSeverity::WARNING);
        if (!strcmp(e, "verbose") || !strcmp(e, "v")) return 50;
        if (!strcmp(e, "debug") || !strcmp(e, "d")) return 40;
        if (!strcmp(e, "info") || !strcmp(e, "i")) return 30;
        if (!strcmp(e, "warning") || !strcmp(e, "w")) return 20;
        if (!strcmp(e, "error") || !strcmp(e, "e")) return 10;
        return atoi(e);
    }();
    return lvl;
}

bool isLoggable(int severity) { return severity <= logLevel(); }

void log(const char* file, int line, int severity, const char* format, ...) {
    const char* tag = severity <= 0 ? "F" : severity <= 10 ? "E" : severity <= 20 ? "W" : severity <= 30 ? "I" : severity <= 40 ? "D" : "V";
    const char* base = strrchr(file, '/');
    fprintf(stderr, "[SNN %s] %s:%d: ", tag, base ? base + 1 : file, line);
    va_list ap;
    va_start(ap, format);
    vfprintf(stderr, format, ap);
    va_end(ap);
    fputc('\n', stderr);
}

void rip() {
    fflush(stderr);
    abort();
}

std::string formatString(const char* format, ...) {
    va_list ap;
    va_start(ap, format);
    va_list ap2;
    va_copy(ap2, ap);
    int n = vsnprintf(nullptr, 0, format, ap);
    va_end(ap);
    std::string s(n > 0 ? n : 0, '\0');
    if (n > 0) vsnprintf(&s[0], static_cast<size_t>(n) + 1, format, ap2);
    va_end(ap2);
    return s;
}

// Restated from the reference's behaviour (core/src/utils.cpp:127-174): keep sign and the top 10 mantissa bits, flush what
// is below the fp16 normal range to signed zero, saturate to infinity above it.  Truncation, not round-to-nearest.
float convertToMediumPrecision(float in) {
    uint32_t u;
    memcpy(&u, &in, 4);
    const uint32_t sign = u & 0x80000000u;
    const int e = static_cast<int>((u >> 23) & 0xFF) - 127 + 15;
    uint32_t r;
    if (e >= 31) {
        r = sign | 0x7F800000u;
    } else if (e <= 0) {
        r = sign;
    } else {
        r = u & 0xFFFFE000u;
    }
    float out;
    memcpy(&out, &r, 4);
    return out;
}

const char* outputDir() {
    const char* e = getenv("SNN_OUTPUT_DIR");
    return e ? e : OUTPUT_DIR;
}

} // namespace snn
