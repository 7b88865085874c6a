"""Runs a chain-shaped net (models.py layer lists) through C-ABI plans: one plan per layer, or fused chain plans."""
import numpy as np

from . import capi


def _layer_plan(ctx, layer, shape):
    n, h, w, c = shape
    t = layer["type"]
    if t == "Conv2D":
        assert c == layer["ic"], (c, layer["ic"])
        return capi.conv2d_plan(ctx, n, h, w, layer["w"], layer["b"], stride=layer["stride"],
                                pads=capi.same_padding(layer["kernel"]) if layer["padding"] == "same" else (0, 0, 0, 0),
                                pad_mode=layer.get("pad_mode", "constant"), act=layer["activation"], leaky=layer.get("alpha", 0.0), bn=layer["bn"])
    if t == "DepthwiseConv2D":
        return capi.conv2d_plan(ctx, n, h, w, layer["w"], layer["b"], stride=layer["stride"],
                                pads=capi.same_padding(layer["kernel"]) if layer["padding"] == "same" else (0, 0, 0, 0), act=layer["activation"],
                                leaky=layer.get("alpha", 0.0), bn=layer["bn"], depthwise=True)
    if t == "Dense":
        return capi.dense_plan(ctx, n, layer["w"], layer["units"], layer["b"], act=layer["activation"] if layer["activation"] in capi.DENSE_ACT else "relu")
    if t == "Subpixel":
        return capi.subpixel_plan(ctx, n, h, w, c, 2, layer.get("mode", 0))
    raise ValueError("unsupported layer type " + t)


class ChainRunner:
    """Pre-builds plans and activations for a fixed input shape; run() only enqueues kernels."""

    def __init__(self, ctx, net, n, h, w, fused=False):
        self.ctx = ctx
        self.net = net
        shape = (n, h, w, net["input_channels"])
        self.in_shape = shape
        self.layer_plans = []
        for layer in net["layers"]:
            p = _layer_plan(ctx, layer, shape)
            self.layer_plans.append(p)
            shape = p.out_shape()
        self.out_shape = shape
        self.x = capi.Tensor(ctx, *self.in_shape)
        if fused:
            self.plans = [capi.chain_plan(ctx, self.layer_plans)]
            self.acts = [capi.Tensor(ctx, *self.out_shape)]
        else:
            self.plans = self.layer_plans
            self.acts = [capi.Tensor(ctx, *p.out_shape()) for p in self.plans]
        self.y = self.acts[-1]

    def describe(self):
        return [p.describe() for p in self.plans]

    def cost(self):
        """(flops, bytes) of the unfused per-layer accounting."""
        f = b = 0.0
        for p in self.layer_plans:
            pf, pb = p.cost()
            f += pf
            b += pb
        return f, b

    def run_device(self):
        src = self.x
        for p, dst in zip(self.plans, self.acts):
            p.run(src, dst)
            src = dst

    def __call__(self, x):
        self.x.upload(np.ascontiguousarray(x, dtype=np.float32))
        self.run_device()
        return self.y.numpy()

    def layer_outputs(self):
        return [a.numpy() for a in self.acts]


class EspcnRunner(ChainRunner):
    pass
