// ic2/layerFactory.h -- name -> creator registry (reference core/src/ic2/layerFactory.{h,cpp}).
#pragma once
#include <string>

#include "ic2/genericlayer.h"

namespace snn {
namespace dp {
typedef GenericModelLayer* (*LayerCreator)(ModelParser& parser, int i, bool useVulkan);
void initLayerRegisty(); // (sic) reference spelling, layerFactory.cpp:109
void registerLayer(const std::string& layerName, LayerCreator creator);
GenericModelLayer* createLayerInstance(std::string layerName, ModelParser& parser, int i, bool useVulkan);
// <Op>Creator1(desc&&, useVulkan): what the reference's unit-test harness calls (layerFactory.h:125-149, shaderUnitTest.cpp:43-44)
GenericModelLayer* Conv2DCreator1(Conv2DDesc&& desc, bool useVulkan);
GenericModelLayer* SeparableConv2DCreator1(SeparableConv2DDesc&& desc, bool useVulkan);
GenericModelLayer* DenseCreator1(DenseDesc&& desc, bool useVulkan);
GenericModelLayer* SubpixelCreator1(SubpixelDesc&& desc, bool useVulkan);
} // namespace dp
} // namespace snn
