// snn/deviceTimer.h -- reference core/inc/snn/deviceTimer.h:20-51 (GPU timestamp pair per stage); HIP flavour = hipEvent pair.
#pragma once
#include <cstdint>
#include <string>
struct snnhip_timer;
struct snnhip_ctx;
namespace snn {
class DeviceTimer {
public:
    explicit DeviceTimer(const std::string& n) : name(n) {}
    virtual ~DeviceTimer() = default;
    virtual void start() = 0;
    virtual void stop() = 0;
    virtual void getTime() = 0;            // resolves the query; duration() is valid afterwards
    uint64_t duration() const { return durationNs; } // nanoseconds, like the reference
    const std::string& getName() const { return name; }

protected:
    std::string name;
    uint64_t durationNs = 0;
};
class HipDeviceTimer : public DeviceTimer {
public:
    HipDeviceTimer(snnhip_ctx* ctx, const std::string& n);
    ~HipDeviceTimer() override;
    void start() override;
    void stop() override;
    void getTime() override;

private:
    snnhip_timer* t = nullptr;
    bool armed = false;
};
} // namespace snn
