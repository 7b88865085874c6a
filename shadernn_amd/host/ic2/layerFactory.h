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
GenericModelLayer* AddCreator1(AddDesc&& desc, bool useVulkan);
GenericModelLayer* ActivationCreator1(ActivationDesc&& desc, bool useVulkan);
GenericModelLayer* BatchNormalizationCreator1(BatchNormalizationDesc&& desc, bool useVulkan);
GenericModelLayer* MaxPooling2DCreator1(MaxPooling2DDesc&& desc, bool useVulkan);
GenericModelLayer* AveragePooling2DCreator1(AveragePooling2DDesc&& desc, bool useVulkan);
GenericModelLayer* AdaptiveAvgPool2dCreator1(AdaptiveAvgPool2dDesc&& desc, bool useVulkan);
GenericModelLayer* FlattenCreator1(FlattenDesc&& desc, bool useVulkan);
GenericModelLayer* PadCreator1(PadDesc&& desc, bool useVulkan);
GenericModelLayer* InstanceNormCreator1(InstanceNormDesc&& desc, bool useVulkan);
GenericModelLayer* UpSampling2DCreator1(UpSampling2DDesc&& desc, bool useVulkan);
GenericModelLayer* ConcatenateCreator1(ConcatenateDesc&& desc, bool useVulkan);
GenericModelLayer* UnaryCreator1(UnaryDesc&& desc, bool useVulkan);
GenericModelLayer* CalculateCreator1(CalculateDesc&& desc, bool useVulkan);
GenericModelLayer* Conv2DTransposeCreator1(Conv2DTransposeDesc&& desc, bool useVulkan);
} // namespace dp
} // namespace snn
