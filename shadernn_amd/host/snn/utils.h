// snn/utils.h -- logging / assertion / timer subset of the reference's core/inc/snn/utils.h (names and behaviour kept:
// SNN_LOG{E,W,I,D,V} controlled by env SNN_LOG_LEVEL, SNN_RIP = log + abort, SNN_CHK, SNN_ASSERT; utils.h:42-98,513-612).
#pragma once
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace snn {

enum class LogSeverity { FATAL = 0, ERROR = 10, WARNING = 20, INFO = 30, DEBUG = 40, VERBOSE = 50 };

bool isLoggable(int severity);
void log(const char* file, int line, int severity, const char* format, ...);
[[noreturn]] void rip();
std::string formatString(const char* format, ...);

// snn::convertToMediumPrecision (reference core/src/utils.cpp:127-174): fp32 -> fp16 by truncation -> fp32
float convertToMediumPrecision(float in);

class Timer {
public:
    explicit Timer(const std::string& n) : name(n) {}
    void start() { t0 = std::chrono::high_resolution_clock::now(); }
    void stop() {
        auto d = std::chrono::high_resolution_clock::now() - t0;
        last = std::chrono::duration_cast<std::chrono::nanoseconds>(d).count();
        total += last;
        ++count;
    }
    uint64_t duration() const { return last; }
    std::string name;
    uint64_t last = 0, total = 0, count = 0;

private:
    std::chrono::high_resolution_clock::time_point t0;
};

} // namespace snn

#define SNN_LOG(sev, ...)                                                                         \
    do {                                                                                          \
        if (::snn::isLoggable(static_cast<int>(::snn::LogSeverity::sev))) ::snn::log(__FILE__, __LINE__, static_cast<int>(::snn::LogSeverity::sev), __VA_ARGS__); \
    } while (0)
#define SNN_LOGE(...) SNN_LOG(ERROR, __VA_ARGS__)
#define SNN_LOGW(...) SNN_LOG(WARNING, __VA_ARGS__)
#define SNN_LOGI(...) SNN_LOG(INFO, __VA_ARGS__)
#define SNN_LOGD(...) SNN_LOG(DEBUG, __VA_ARGS__)
#define SNN_LOGV(...) SNN_LOG(VERBOSE, __VA_ARGS__)
#define SNN_RIP(...)                                                                       \
    do {                                                                                   \
        ::snn::log(__FILE__, __LINE__, static_cast<int>(::snn::LogSeverity::FATAL), __VA_ARGS__); \
        ::snn::rip();                                                                      \
    } while (0)
#define SNN_CHK(x)                        \
    do {                                  \
        if (!(x)) SNN_RIP("%s", #x);      \
    } while (0)
#ifdef NDEBUG
#define SNN_ASSERT(x) ((void) 0)
#else
#define SNN_ASSERT(x) SNN_CHK(x)
#endif
#define SNN_NO_COPY(X) \
    X(const X&) = delete; \
    X& operator=(const X&) = delete
#define SNN_NO_MOVE(X) \
    X(X&&) = delete;   \
    X& operator=(X&&) = delete
