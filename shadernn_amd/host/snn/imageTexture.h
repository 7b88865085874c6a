// snn/imageTexture.h -- "an image located on GPU or CPU" (reference core/inc/snn/imageTexture.h:31-372).
// The reference stores RGBA 3-D textures ([ceil(C/4)][H][W][4], "C4HW4"); the HIP flavour keeps a plain NHWC fp32 tensor in
// HBM (true channel count) and converts to/from C4HW4 only in upload()/download() so that dumps stay byte compatible.
#pragma once
#include <array>
#include <memory>
#include <string>
#include <vector>

#include "snn/image.h"
#include "snn/snn.h"

struct snnhip_tensor;
struct snnhip_ctx;

namespace snn {

typedef enum class Backend { Backend_CPU, Backend_GPU, NOT_DEFINED = 200 } Backend;

class ImageTexture {
public:
    explicit ImageTexture(GpuContext* context) : _context(context) {}
    virtual ~ImageTexture();
    SNN_NO_COPY(ImageTexture);

    // dims = {width, height, depth (= ceil(channels/4) texels), planes}; channels = true channel count (0 => 4*depth)
    void reset(const std::array<uint32_t, 4>& dims, ColorFormat format, const void* buffer = nullptr, const std::string& name = "", uint32_t channels = 0);
    virtual void resetTexture(const std::array<uint32_t, 4>& dims, ColorFormat format, const std::string& name = "", uint32_t channels = 0);
    virtual void attach(ImageTexture* src); // alias the producer's device tensor (core.cpp:361)
    virtual bool isValid() const { return _tensor != nullptr; }
    virtual void upload();   // host C4HW4 image -> HBM NHWC
    virtual void download(); // HBM NHWC -> host C4HW4 image
    void uploadNHWC(const float* nhwc);
    void downloadNHWC(float* nhwc);

    // Pre-processing either side of a model run, on the device:
    //  loadU8AndNormalize = loadFromFile's decoded 8-bit pixels + convertToRGBA32FAndNormalize (imageTexture.h:114, image.cpp:712-796): an
    //      R8 / RGB8 / RGBA8 image becomes this texture's 4-channel tensor, y = (u8 - means) * norms (RGB8 alpha = 1, R8 see norm2rgba32f);
    //  resize = ImageTextureVulkan::resize (imageTextureVulkan.cpp:137-183): the tensor is replaced by one of round(w / xScale) x
    //      round(h / yScale), each texel (sample - means[c % 4]) * norms[c % 4], bilinear (clamp to edge) or nearest.
    void loadU8AndNormalize(const uint8_t* pixels, uint32_t w, uint32_t h, uint32_t srcChannels, const std::array<float, 4>& means = {{0, 0, 0, 0}},
                            const std::array<float, 4>& norms = {{1, 1, 1, 1}});
    bool resize(float xScale, float yScale, const std::array<float, 4>& means, const std::array<float, 4>& norms, bool linearFilter = true);

    const std::array<uint32_t, 4>& getDims() const { return _dims; }
    uint32_t width() const { return _dims[0]; }
    uint32_t height() const { return _dims[1]; }
    uint32_t depth() const { return _dims[2]; }
    uint32_t channels() const { return _channels; }
    uint32_t batch() const { return _dims[3] ? _dims[3] : 1; } // HIP extension: dims[3] (the reference's unused "planes" slot, always 1 there) = images
    ColorFormat getFormat() const { return _format; }
    const std::string& getName() const { return _name; }
    std::string getTextureInfo2() const;
    RawImage& image() { return _image; }
    void saveToBIN(const std::string& filename); // download + RawImage::saveToBIN (imageTexture.h:325-331)
    snnhip_tensor* tensor() const { return _tensor; }

    // CPU-layer hand-off used by the reference's Dense/Flatten path (imageTexture.h:349)
    std::vector<std::vector<float>>& getOutputMat() { return outputMat; }
    void setOutputMat(const std::vector<std::vector<float>>& m) { outputMat = m; }

protected:
    GpuContext* _context;
    std::string _name;
    ColorFormat _format = ColorFormat::NONE;
    std::array<uint32_t, 4> _dims{{0, 0, 0, 0}};
    uint32_t _channels = 0;
    RawImage _image;
    snnhip_tensor* _tensor = nullptr;
    bool _ownsTensor = false;
    std::vector<std::vector<float>> outputMat;
    snnhip_ctx* hipCtx() const;
    void releaseTensor();
};

// The reference uses PolyArray<ImageTexture,...> (utils.h); a vector of shared pointers gives the same call shapes:
// arr[i].attach(...), arr.size(), arr.allocate(n).
class ImageTextureArray {
public:
    explicit ImageTextureArray(GpuContext* c = nullptr) : context(c) {}
    void allocate(size_t n);
    size_t size() const { return items.size(); }
    ImageTexture& operator[](size_t i) { return *items[i]; }
    const ImageTexture& operator[](size_t i) const { return *items[i]; }
    std::shared_ptr<ImageTexture>& ptr(size_t i) { return items[i]; }
    void push_back(std::shared_ptr<ImageTexture> t) { items.push_back(std::move(t)); }
    GpuContext* context;

private:
    std::vector<std::shared_ptr<ImageTexture>> items;
};
typedef ImageTextureArray& ImageTextureArrayAccessor;

struct ImageTextureFactory {
    // imageTextureFactory.cpp:27-85
    static std::shared_ptr<ImageTexture> createImageTexture(GpuContext* context, const std::array<uint32_t, 4>& dims, ColorFormat format,
                                                            const void* buffer = nullptr, uint32_t channels = 0);
};

} // namespace snn
