// image.cpp -- RawImage + the ".dump" raw format (reference core/src/image.cpp:216-245, 274-311).
#include <cstring>
#include <fstream>
#include <sstream>

#include "snn/image.h"
#include "snn/utils.h"

namespace snn {

RawImage::RawImage(const ImageDesc& d, const void* pixels) : _desc(d) {
    const size_t bytes = static_cast<size_t>(d.width) * d.height * d.depth * getColorFormatDesc(d.format).bytes();
    _pixels.assign(bytes, 0);
    if (pixels) memcpy(_pixels.data(), pixels, bytes);
}

void RawImage::saveToBIN(const std::string& filepath) const {
    std::ofstream fp(filepath, std::ios::binary);
    if (!fp.good()) {
        SNN_LOGE("open %s failed", filepath.c_str());
        return;
    }
    char header[32] = {};
    std::snprintf(header, 32, "%d %d %d %d", width(), height(), depth(), channels());
    fp.write(header, 32);
    fp.write(reinterpret_cast<const char*>(_pixels.data()), static_cast<std::streamsize>(_pixels.size()));
}

RawImage RawImage::loadFromBIN(const std::string& filename) {
    std::ifstream file(filename, std::ios::binary);
    if (!file.good()) SNN_RIP("Failed to open image file %s", filename.c_str());
    std::vector<char> buf((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
    if (buf.size() < 32) SNN_RIP("File %s has incorrect bin format !", filename.c_str());
    int w = -1, h = -1, d = -1, c = -1;
    std::stringstream header(std::string(buf.data(), 32));
    header >> w >> h >> d >> c;
    if (w < 0 || h < 0 || d <= 0 || c < 0 || c % d != 0) SNN_RIP("File %s has incorrect bin header format !", filename.c_str());
    const int comps = c / d;
    ImageDesc desc;
    desc.format = comps == 4 ? ColorFormat::RGBA32F : ColorFormat::R32F;
    desc.width = static_cast<uint32_t>(w);
    desc.height = static_cast<uint32_t>(h);
    desc.depth = static_cast<uint32_t>(d);
    desc.channels = static_cast<uint32_t>(c);
    const size_t want = static_cast<size_t>(w) * h * d * getColorFormatDesc(desc.format).bytes();
    if (buf.size() - 32 < want) SNN_RIP("File %s is truncated", filename.c_str());
    return RawImage(desc, buf.data() + 32);
}

} // namespace snn
