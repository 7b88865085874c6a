// snn/image.h -- host-side image container + the reference's raw ".dump" format
// (core/src/image.cpp:216-245: 32-byte ASCII header "W H D C" NUL padded, then raw pixels; RGBA32F = [D][H][W][4]).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "snn/color.h"

namespace snn {
struct ImageDesc {
    ColorFormat format = ColorFormat::NONE;
    uint32_t width = 0, height = 0, depth = 0, channels = 0; // channels = total channel count (4*depth for RGBA planes)
};
class RawImage {
public:
    RawImage() = default;
    RawImage(const ImageDesc& d, const void* pixels = nullptr);
    const ImageDesc& desc() const { return _desc; }
    uint32_t width() const { return _desc.width; }
    uint32_t height() const { return _desc.height; }
    uint32_t depth() const { return _desc.depth; }
    uint32_t channels() const { return _desc.channels; }
    ColorFormat format() const { return _desc.format; }
    size_t size() const { return _pixels.size(); }
    const uint8_t* data() const { return _pixels.data(); }
    uint8_t* data() { return _pixels.data(); }
    bool empty() const { return _pixels.empty(); }
    void saveToBIN(const std::string& path) const;               // image.cpp:216-245
    static RawImage loadFromBIN(const std::string& path);        // image.cpp:300-311 (header parse)

private:
    ImageDesc _desc;
    std::vector<uint8_t> _pixels;
};
} // namespace snn
