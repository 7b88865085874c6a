// snn/color.h -- the colour formats the hot path uses (reference core/inc/snn/color.h:20-110).
#pragma once
#include <cstddef>
namespace snn {
enum class ColorFormat { NONE, RGBA32F, RGBA16F, R32F };
struct ColorFormatDesc {
    const char* name;
    size_t bits, ch;
    size_t bytes() const { return bits / 8U; }
};
inline ColorFormatDesc getColorFormatDesc(ColorFormat f) {
    switch (f) {
    case ColorFormat::RGBA32F: return {"RGBA32F", 128, 4};
    case ColorFormat::RGBA16F: return {"RGBA16F", 64, 4};
    case ColorFormat::R32F: return {"R32F", 32, 1};
    default: return {"NONE", 0, 0};
    }
}
} // namespace snn
