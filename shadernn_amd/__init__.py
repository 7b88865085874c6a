"""shadernn_amd -- MI355X-native backend for ShaderNN's conv / depthwise / dense operator hot path.

Python here is test/bench plumbing over the C-ABI (include/snnhip.h -> lib/libsnnhip.so) and over the C++ host mirror
of the reference API (lib/libsnn_core.so).  There is no CPU fallback anywhere in this package.
"""
from .capi import (ACT, DENSE_ACT, F16, F32, U8, PAD_MODE, Context, Graph, Plan, SnnHipError, Tensor, Timer, activation_plan, add_plan, batchnorm_plan, chain_plan,  # noqa: F401
                   conv2d_plan, dense_plan, global_avgpool_plan, instancenorm_plan, lib, load_library, pad_plan, pool2d_plan, same_padding,
                   subpixel_plan, upsample_plan, concat_plan, unary_plan, calculate_plan, resize_plan, image_u8_plan, deconv2d_plan, graph_fuse, E_UNSUPPORTED, set_option, get_option)
from .runner import ChainRunner, EspcnRunner, GraphRunner  # noqa: F401
