// snn/contextFactory.h -- reference core/inc/snn/contextFactory.h + core/src/contextFactory.cpp:21-32.
#pragma once
#include "snn/snn.h"
struct snnhip_ctx;
namespace snn {
// GpuContext flavour owning a snnhip_ctx (one per device; multi-GPU = one context per process/device).
class HipContext : public GpuContext {
public:
    explicit HipContext(int device = 0);
    ~HipContext() override;
    snnhip_ctx* ctx = nullptr;
    int device = 0;
};
// createDefaultContext(useVulkan) of the reference picks GL or Vulkan; here the only backend is HIP.
GpuContext* createDefaultContext(bool useVulkan = false);
GpuContext* createHipContext(int device);
} // namespace snn
