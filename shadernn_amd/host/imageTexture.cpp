// imageTexture.cpp -- HIP flavour of ImageTexture, the context factory and the device timer
// (reference core/src/imageTexture*.cpp, imageTextureFactory.cpp:27-85, contextFactory.cpp:21-32, vkUtils timers).
#include <cmath>
#include <cstring>

#include "../../include/snnhip.h"
#include "snn/contextFactory.h"
#include "snn/deviceTimer.h"
#include "snn/imageTexture.h"

namespace snn {

static void hipChk(int rc, const char* what) {
    if (rc != SNNHIP_OK) SNN_RIP("%s: %s", what, snnhip_last_error()); // errors are fatal, like BM_CHECK_OK / SNN_RIP in the reference
}

// ---- context
HipContext::HipContext(int dev) : GpuContext(GpuBackendType::HIP), device(dev) { hipChk(snnhip_ctx_create(dev, &ctx), "snnhip_ctx_create"); }
HipContext::~HipContext() { snnhip_ctx_destroy(ctx); }
GpuContext* createHipContext(int device) { return new HipContext(device); }
GpuContext* createDefaultContext(bool) {
    const char* e = getenv("SNN_HIP_DEVICE");
    return new HipContext(e ? atoi(e) : 0);
}

// ---- timer
HipDeviceTimer::HipDeviceTimer(snnhip_ctx* ctx, const std::string& n) : DeviceTimer(n) { hipChk(snnhip_timer_create(ctx, &t), "snnhip_timer_create"); }
HipDeviceTimer::~HipDeviceTimer() { snnhip_timer_destroy(t); }
void HipDeviceTimer::start() {
    hipChk(snnhip_timer_start(t), "snnhip_timer_start");
    armed = false;
}
void HipDeviceTimer::stop() {
    hipChk(snnhip_timer_stop(t), "snnhip_timer_stop");
    armed = true;
}
void HipDeviceTimer::getTime() {
    if (!armed) return;
    float ms = 0;
    hipChk(snnhip_timer_elapsed_ms(t, &ms), "snnhip_timer_elapsed_ms");
    durationNs = static_cast<uint64_t>(static_cast<double>(ms) * 1e6);
    armed = false;
}

// ---- ImageTexture
snnhip_ctx* ImageTexture::hipCtx() const {
    SNN_CHK(_context && _context->backendType == GpuBackendType::HIP);
    return static_cast<HipContext*>(_context)->ctx;
}

ImageTexture::~ImageTexture() { releaseTensor(); }

void ImageTexture::releaseTensor() {
    if (_tensor && _ownsTensor) snnhip_tensor_free(_tensor);
    _tensor = nullptr;
    _ownsTensor = false;
}

void ImageTexture::reset(const std::array<uint32_t, 4>& dims, ColorFormat format, const void* buffer, const std::string& name, uint32_t channels) {
    _dims = dims;
    _format = format;
    _name = name;
    _channels = channels ? channels : 4 * dims[2];
    if (_channels > 4 * dims[2]) SNN_RIP("texture with %u channels needs more than %u texel planes", _channels, dims[2]);
    ImageDesc d;
    d.format = ColorFormat::RGBA32F; // host staging is always fp32; an RGBA16F texture is a half tensor in HBM and converts at the C-ABI edge
    d.width = dims[0];
    d.height = dims[1];
    d.depth = dims[2] * batch(); // a batched texture stages (and dumps) its images as consecutive groups of depth texel planes
    d.channels = 4 * d.depth;
    _image = RawImage(d, buffer);
}

void ImageTexture::resetTexture(const std::array<uint32_t, 4>& dims, ColorFormat format, const std::string& name, uint32_t channels) {
    if (format != ColorFormat::RGBA32F && format != ColorFormat::RGBA16F)
        SNN_RIP("HIP backend: RGBA32F (fp32) and RGBA16F (fp16) textures are implemented, got %s", getColorFormatDesc(format).name);
    releaseTensor();
    reset(dims, format, nullptr, name, channels);
    hipChk(snnhip_tensor_alloc(hipCtx(), static_cast<int>(batch()), static_cast<int>(dims[1]), static_cast<int>(dims[0]), static_cast<int>(_channels),
                               format == ColorFormat::RGBA16F ? SNNHIP_F16 : SNNHIP_F32, &_tensor),
           "snnhip_tensor_alloc");
    _ownsTensor = true;
}

void ImageTexture::attach(ImageTexture* src) { // consumer input aliases the producer's output (core.cpp:361)
    releaseTensor();
    _tensor = src->_tensor;
    _ownsTensor = false;
    _dims = src->_dims;
    _format = src->_format;
    _channels = src->_channels;
}

void ImageTexture::upload() {
    SNN_CHK(_tensor && !_image.empty());
    hipChk(snnhip_tensor_upload_c4hw4(_tensor, reinterpret_cast<const float*>(_image.data())), "snnhip_tensor_upload_c4hw4");
}

void ImageTexture::download() {
    SNN_CHK(_tensor);
    if (_image.empty() || _image.width() != _dims[0] || _image.height() != _dims[1] || _image.depth() != _dims[2] * batch()) {
        ImageDesc d;
        d.format = ColorFormat::RGBA32F;
        d.width = _dims[0];
        d.height = _dims[1];
        d.depth = _dims[2] * batch();
        d.channels = 4 * d.depth;
        _image = RawImage(d, nullptr);
    }
    hipChk(snnhip_tensor_download_c4hw4(_tensor, reinterpret_cast<float*>(_image.data())), "snnhip_tensor_download_c4hw4");
}

void ImageTexture::uploadNHWC(const float* nhwc) {
    SNN_CHK(_tensor);
    hipChk(snnhip_tensor_upload(_tensor, nhwc), "snnhip_tensor_upload");
}

void ImageTexture::downloadNHWC(float* nhwc) {
    SNN_CHK(_tensor);
    hipChk(snnhip_tensor_download(_tensor, nhwc), "snnhip_tensor_download");
}

void ImageTexture::loadU8AndNormalize(const uint8_t* pixels, uint32_t w, uint32_t h, uint32_t srcChannels, const std::array<float, 4>& means,
                                      const std::array<float, 4>& norms) {
    SNN_CHK(pixels && (srcChannels == 1 || srcChannels == 3 || srcChannels == 4));
    if (batch() != 1) SNN_RIP("loadU8AndNormalize: one image per call, this texture holds a batch of %u", batch());
    const ColorFormat fmt = _format == ColorFormat::RGBA16F ? ColorFormat::RGBA16F : ColorFormat::RGBA32F;
    resetTexture({w, h, 1, 1}, fmt, _name, 4);
    snnhip_tensor* src = nullptr;
    hipChk(snnhip_tensor_alloc(hipCtx(), 1, static_cast<int>(h), static_cast<int>(w), static_cast<int>(srcChannels), SNNHIP_U8, &src), "snnhip_tensor_alloc(u8)");
    hipChk(snnhip_tensor_upload_raw(src, pixels, static_cast<size_t>(w) * h * srcChannels), "snnhip_tensor_upload_raw");
    snnhip_image_u8_desc d = {};
    d.N = 1;
    d.H = static_cast<int>(h);
    d.W = static_cast<int>(w);
    d.src_channels = static_cast<int>(srcChannels);
    for (int i = 0; i < 4; ++i) {
        d.means[i] = means[static_cast<size_t>(i)];
        d.norms[i] = norms[static_cast<size_t>(i)];
    }
    snnhip_plan* plan = nullptr;
    hipChk(snnhip_image_u8_plan_create(hipCtx(), &d, &plan), "snnhip_image_u8_plan_create");
    hipChk(snnhip_plan_run(plan, src, _tensor), "snnhip_plan_run(image_u8)");
    hipChk(snnhip_sync(hipCtx()), "snnhip_sync"); // src is freed below
    snnhip_plan_destroy(plan);
    snnhip_tensor_free(src);
}

bool ImageTexture::resize(float xScale, float yScale, const std::array<float, 4>& means, const std::array<float, 4>& norms, bool linearFilter) {
    SNN_CHK(_tensor && xScale > 0.0f && yScale > 0.0f);
    if (batch() != 1) SNN_RIP("resize: one image per call, this texture holds a batch of %u", batch());
    snnhip_resize_desc d = {};
    d.N = 1;
    d.H = static_cast<int>(_dims[1]);
    d.W = static_cast<int>(_dims[0]);
    d.C = static_cast<int>(_channels);
    d.OW = static_cast<int>(roundf(static_cast<float>(_dims[0]) / xScale)); // imageTextureVulkan.cpp:150-151
    d.OH = static_cast<int>(roundf(static_cast<float>(_dims[1]) / yScale));
    d.linear = linearFilter ? 1 : 0;
    for (int i = 0; i < 4; ++i) {
        d.means[i] = means[static_cast<size_t>(i)];
        d.norms[i] = norms[static_cast<size_t>(i)];
    }
    snnhip_tensor* dst = nullptr;
    hipChk(snnhip_tensor_alloc(hipCtx(), 1, d.OH, d.OW, d.C, _format == ColorFormat::RGBA16F ? SNNHIP_F16 : SNNHIP_F32, &dst), "snnhip_tensor_alloc");
    snnhip_plan* plan = nullptr;
    hipChk(snnhip_resize_plan_create(hipCtx(), &d, &plan), "snnhip_resize_plan_create");
    hipChk(snnhip_plan_run(plan, _tensor, dst), "snnhip_plan_run(resize)");
    hipChk(snnhip_sync(hipCtx()), "snnhip_sync");
    snnhip_plan_destroy(plan);
    releaseTensor(); // the resized image replaces the source (imageTextureVulkan.cpp:176-178)
    _tensor = dst;
    _ownsTensor = true;
    _dims[0] = static_cast<uint32_t>(d.OW);
    _dims[1] = static_cast<uint32_t>(d.OH);
    _image = RawImage();
    return false; // the reference returns 0 on success
}

std::string ImageTexture::getTextureInfo2() const {
    return formatString("%s %ux%ux%u (C=%u, batch %u) %s tensor=%p", _name.c_str(), _dims[0], _dims[1], _dims[2], _channels, batch(),
                        getColorFormatDesc(_format).name, static_cast<void*>(_tensor));
}

void ImageTexture::saveToBIN(const std::string& filename) {
    download();
    _image.saveToBIN(filename);
}

void ImageTextureArray::allocate(size_t n) {
    items.clear();
    for (size_t i = 0; i < n; ++i) items.push_back(std::make_shared<ImageTexture>(context));
}

std::shared_ptr<ImageTexture> ImageTextureFactory::createImageTexture(GpuContext* context, const std::array<uint32_t, 4>& dims, ColorFormat format,
                                                                      const void* buffer, uint32_t channels) {
    auto t = std::make_shared<ImageTexture>(context);
    t->resetTexture(dims, format, "", channels);
    if (buffer) {
        memcpy(t->image().data(), buffer, t->image().size());
    }
    return t;
}

} // namespace snn
