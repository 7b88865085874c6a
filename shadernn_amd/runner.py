"""Runs a chain-shaped net (models.py layer lists) through C-ABI plans: one plan per layer, or fused chain plans."""
import numpy as np

from . import capi


def _layer_plan(ctx, layer, shape, dtype=capi.F32):
    n, h, w, c = shape
    t = layer["type"]
    if t == "Conv2D":
        assert c == layer["ic"], (c, layer["ic"])
        return capi.conv2d_plan(ctx, n, h, w, layer["w"], layer["b"], stride=layer["stride"],
                                pads=capi.same_padding(layer["kernel"]) if layer["padding"] == "same" else (0, 0, 0, 0),
                                pad_mode=layer.get("pad_mode", "constant"), act=layer["activation"], leaky=layer.get("alpha", 0.0), bn=layer["bn"], dtype=dtype)
    if t == "DepthwiseConv2D":
        return capi.conv2d_plan(ctx, n, h, w, layer["w"], layer["b"], stride=layer["stride"],
                                pads=capi.same_padding(layer["kernel"]) if layer["padding"] == "same" else (0, 0, 0, 0), act=layer["activation"],
                                leaky=layer.get("alpha", 0.0), bn=layer["bn"], depthwise=True, dtype=dtype)
    if t == "Dense":
        return capi.dense_plan(ctx, n, layer["w"], layer["units"], layer["b"], act=layer["activation"] if layer["activation"] in capi.DENSE_ACT else "relu")
    if t == "Subpixel":
        return capi.subpixel_plan(ctx, n, h, w, c, 2, layer.get("mode", 0))
    if t in ("MaxPooling2D", "AveragePooling2D"):
        return capi.pool2d_plan(ctx, n, h, w, c, layer["pool"], layer["stride"], kind="max" if t == "MaxPooling2D" else "avg",
                                same=layer["padding"] not in ("valid", "none", "0"))
    if t == "AdaptiveAvgPool2d":
        return capi.global_avgpool_plan(ctx, n, h, w, c)
    if t == "Add":
        return capi.add_plan(ctx, n, h, w, c, act=_plain(layer.get("activation", "")), leaky=layer.get("alpha", 0.0))
    if t in ("Activation", "Flatten"):
        return capi.activation_plan(ctx, n, h, w, c, _plain(layer.get("activation", "")), layer.get("alpha", 0.0))
    if t == "BatchNormalization":
        return capi.batchnorm_plan(ctx, n, h, w, c, layer["bn"], act=_plain(layer.get("activation", "")), leaky=layer.get("alpha", 0.0))
    if t == "Pad":
        (pt, pb), (pl, pr) = layer["padding"]
        return capi.pad_plan(ctx, n, h, w, c, (pt, pb, pl, pr), layer["mode"])
    if t == "InstanceNorm":
        return capi.instancenorm_plan(ctx, n, h, w, c, layer["beta"], layer["gamma"], act=_plain(layer.get("activation", "")), leaky=layer.get("alpha", 0.0))
    if t == "UpSampling2D":
        return capi.upsample_plan(ctx, n, h, w, c, layer["scaleFactor"], layer["interpolation"])
    if t == "Concatenate":
        return capi.concat_plan(ctx, n, h, w, layer["c0"], layer["c1"], layer.get("oc"))
    if t == "Calculate":
        return capi.calculate_plan(ctx, n, h, w, c, layer["oc"])
    if t == "Unary":
        return capi.unary_plan(ctx, n, h, w, c, layer.get("op", "copy"), layer.get("value", 1.0))
    if t == "Conv2DTranspose":
        return capi.deconv2d_plan(ctx, n, h, w, layer["w"], layer["b"], stride=layer["stride"], same=layer["padding"] == "same",
                                  act=_plain(layer["activation"]), leaky=layer.get("alpha", 0.0), bn=layer["bn"])
    raise ValueError("unsupported layer type " + t)


def _plain(act):
    return "" if act in ("linear", "none", None) else act


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


class GraphRunner:
    """Graph-shaped nets (models.resnet18 / mobilenetv2 / style_net): one plan per layer at any batch size, producers by name.
    Tensors are allocated once; run_device() only enqueues kernels.  (The C++ host mirror runs the same graphs from JSON.)"""

    def __init__(self, ctx, net, n, h, w, dtype=capi.F32, fuse=True):
        """dtype=capi.F16: half tensors end to end (convolutions on the fp16 MFMA path, fp32 accumulation).
        fuse=True: the layer DAG goes through snnhip_graph_fuse -- the same single fusion pass the C++ host's HipBackend::finalizeStages
        uses (residual Conv2D -> Add pairs, [UpSampling2D ->] Pad -> Conv2D strings, ...).  Layers folded into a fused plan produce no
        tensor of their own (output_of() is unavailable for them); `steps` lists what really runs."""
        from . import models

        self.ctx, self.net, self.dtype = ctx, net, dtype
        self.in_shape = (n, h, w, net["input_channels"])
        self.x = capi.Tensor(ctx, *self.in_shape, dtype=dtype)
        shapes = {"input": self.in_shape}
        prods = models.producers(net)
        index = {"input": -1}
        nodes, self.layer_plans = [], []
        self.output_names = models.output_names(net)  # several for multi-head graphs (YOLOv3-tiny)
        for k, (layer, ins) in enumerate(prods):
            shape = shapes[ins[0]]
            if layer["type"] == "Add":  # output extent = max over the inputs (genericlayer.cpp:64-90)
                shape = (shape[0], max(shapes[i][1] for i in ins), max(shapes[i][2] for i in ins), shape[3])
            if layer["type"] == "Dense":  # consumes the flattened producer
                shape = (shape[0], 1, 1, shape[1] * shape[2] * shape[3])
            plan = _layer_plan(ctx, layer, shape, dtype)
            out_shape = plan.out_shape()
            if layer["type"] == "Flatten":
                out_shape = (shape[0], 1, 1, shape[1] * shape[2] * shape[3])
            shapes[layer["name"]] = out_shape
            index[layer["name"]] = k
            self.layer_plans.append(plan)
            nodes.append((plan, [index[i] for i in ins], layer["name"] in self.output_names))
        fused = capi.graph_fuse(ctx, nodes) if fuse else [(p, i) for p, i, _ in nodes]
        names = ["input"] + [l["name"] for l, _ in prods]  # node index + 1
        self.tensors = {"input": self.x}
        self.steps = []  # (plan, [input tensors], output tensor, layer)
        self.fused_layers = []
        for k, ((layer, _), (plan, ins)) in enumerate(zip(prods, fused)):
            if plan is None:
                self.fused_layers.append(layer["name"])
                continue
            t = capi.Tensor(ctx, *shapes[layer["name"]], dtype=dtype)
            self.tensors[layer["name"]] = t
            self.steps.append((plan, [self.tensors[names[i + 1]] for i in ins], t, layer))
        self.fused_norms = [l["name"] for p, _, _, l in self.steps if "instancenorm(fold" in p.describe()]  # chain rule F (opt-in)
        self.y = self.steps[-1][2]
        self.out_shape = shapes[net["layers"][-1]["name"]]

    def describe(self):
        return ["%s: %s" % (l["name"], p.describe()) for p, _, _, l in self.steps]

    def step_cost(self, i):
        """(flops, bytes) of step i; the dtype-agnostic element-wise plans report fp32 bytes, halved here for half tensors."""
        plan, _, _, layer = self.steps[i]
        f, b = plan.cost()
        if self.dtype == capi.F16 and layer["type"] not in ("Conv2D", "DepthwiseConv2D", "Dense", "Conv2DTranspose"):
            b *= 0.5
        return f, b

    def cost(self):
        f = b = 0.0
        for i in range(len(self.steps)):
            pf, pb = self.step_cost(i)
            f += pf
            b += pb
        return f, b

    def run_device(self):
        for plan, ins, out, _ in self.steps:
            plan.run(ins if len(ins) > 1 else ins[0], out)

    def __call__(self, x):
        self.x.upload(np.ascontiguousarray(x, dtype=np.float32))
        self.run_device()
        return self.y.numpy()

    def outputs(self):
        return [self.tensors[n].numpy() for n in self.output_names]

    def output_of(self, name):
        return self.tensors[name].numpy()
