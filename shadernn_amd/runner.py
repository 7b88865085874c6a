"""Runs a chain-shaped net (models.py layer lists) through C-ABI plans: one plan per layer, or fused chain plans."""
import os

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
    Tensors are allocated once; run_device() only enqueues kernels.  (The C++ host mirror runs the same graphs at batch 1 from JSON.)"""

    def __init__(self, ctx, net, n, h, w, dtype=capi.F32, fuse=True):
        """dtype=capi.F16: half tensors end to end (convolutions on the fp16 MFMA path, fp32 accumulation).
        fuse=True: a Pad layer whose only consumer is a Conv2D is folded into that convolution's tile staging through
        snnhip_chain_plan_create (rule D); the pad's own tensor is then never produced (output_of(pad) is unavailable)."""
        from . import models

        self.ctx, self.net, self.dtype = ctx, net, dtype
        self.in_shape = (n, h, w, net["input_channels"])
        self.x = capi.Tensor(ctx, *self.in_shape, dtype=dtype)
        shapes, self.tensors = {"input": self.in_shape}, {"input": self.x}
        self.steps = []  # (plan, [input tensors], output tensor, layer)
        for layer, ins in models.producers(net):
            shape = shapes[ins[0]]
            if layer["type"] == "Add":  # output extent = max over the inputs (genericlayer.cpp:64-90)
                shape = (shape[0], max(shapes[i][1] for i in ins), max(shapes[i][2] for i in ins), shape[3])
            if layer["type"] == "Dense":  # consumes the flattened producer
                shape = (shape[0], 1, 1, shape[1] * shape[2] * shape[3])
            plan = _layer_plan(ctx, layer, shape, dtype)
            out_shape = plan.out_shape()
            if layer["type"] == "Flatten":
                out_shape = (shape[0], 1, 1, shape[1] * shape[2] * shape[3])
            t = capi.Tensor(ctx, *out_shape, dtype=dtype)
            shapes[layer["name"]], self.tensors[layer["name"]] = out_shape, t
            self.steps.append((plan, [self.tensors[i] for i in ins], t, layer))
        self.output_names = models.output_names(net)  # several for multi-head graphs (YOLOv3-tiny)
        self.fused_pads = []
        self.fused_norms = []
        self.fused_adds = []
        if fuse:
            self._fuse_pads(models.producers(net))
            self._fuse_adds(models.producers(net))
        self.y = self.steps[-1][2]
        self.out_shape = shapes[net["layers"][-1]["name"]]

    def _fuse_pads(self, prods):
        """[UpSampling2D(nearest x2) ->] [Pad ->] Conv2D: the convolution stages its tiles straight from the tensor in front of the pad /
        upsampling (chain rule D); the folded layers need a single consumer each."""
        consumers = {}
        for layer, ins in prods:
            for i in ins:
                consumers.setdefault(i, []).append(layer["name"])
        outputs = set(self.output_names)
        name_of = {id(t): nm for nm, t in self.tensors.items()}
        by_name = {l["name"]: k for k, (_, _, _, l) in enumerate(self.steps)}
        drop = set()

        def foldable(nm, types):
            if nm not in by_name or nm in outputs or len(consumers.get(nm, [])) != 1:
                return None
            st = self.steps[by_name[nm]]
            return st if st[3]["type"] in types else None

        for k, (plan, ins, out, layer) in enumerate(self.steps):
            if layer["type"] != "Conv2D":
                continue
            chain_plans, first_ins, folded = [plan], ins, []
            src = name_of.get(id(ins[0]))
            st = foldable(src, ("Pad",))
            if st:
                chain_plans.insert(0, st[0])
                first_ins, folded = st[1], [st[3]["name"]]
                src = name_of.get(id(st[1][0]))
            st = foldable(src, ("UpSampling2D",))
            if st and st[3]["interpolation"] == "nearest" and float(st[3]["scaleFactor"]) == 2.0:
                chain_plans.insert(0, st[0])
                first_ins, folded = st[1], folded + [st[3]["name"]]
            # chain rule F: an InstanceNorm that is the only consumer of this convolution takes its statistics from the convolution's epilogue
            norm = None
            users = consumers.get(layer["name"], [])
            if len(users) == 1 and layer["name"] not in outputs and users[0] in by_name and self.steps[by_name[users[0]]][3]["type"] == "InstanceNorm":
                norm = self.steps[by_name[users[0]]]
            if norm is not None and os.environ.get("SNNHIP_NORM_FUSION", "0") not in ("", "0"):  # opt-in: measured neutral to negative
                try:
                    chain = capi.chain_plan(self.ctx, chain_plans + [norm[0]])
                except capi.SnnHipError as e:
                    if e.code != capi.E_UNSUPPORTED:
                        raise
                    chain = None
                if chain is not None and chain.num_steps() == 1:
                    self.steps[k] = (chain, first_ins, norm[2], norm[3])
                    drop.update(by_name[nm] for nm in folded + [norm[3]["name"]])
                    self.fused_pads += folded
                    self.fused_norms.append(norm[3]["name"])
                    continue
                if chain is not None:
                    chain.destroy()
            while len(chain_plans) > 1:
                try:
                    chain = capi.chain_plan(self.ctx, chain_plans)
                except capi.SnnHipError as e:
                    if e.code != capi.E_UNSUPPORTED:
                        raise
                    chain = None
                if chain is not None and chain.num_steps() == 1:
                    self.steps[k] = (chain, first_ins, out, layer)
                    drop.update(by_name[nm] for nm in folded)
                    self.fused_pads += folded
                    break
                if chain is not None:
                    chain.destroy()
                if len(chain_plans) == 3:  # the upsampling could not be folded: retry with the pad alone
                    chain_plans, folded = chain_plans[1:], folded[:1]
                    first_ins = self.steps[by_name[folded[0]]][1]
                else:
                    break
        self.steps = [st for k, st in enumerate(self.steps) if k not in drop]

    def _fuse_adds(self, prods):
        """Conv2D -> Add (the residual connections of ResNet / MobileNetV2): the add moves into the convolution's epilogue (chain rule E)."""
        consumers = {}
        for layer, ins in prods:
            for i in ins:
                consumers.setdefault(i, []).append(layer["name"])
        outputs = set(self.output_names)
        name_of = {id(t): nm for nm, t in self.tensors.items()}
        by_name = {l["name"]: k for k, (_, _, _, l) in enumerate(self.steps)}
        drop, replace = set(), {}
        for k, (plan, ins, out, layer) in enumerate(self.steps):
            if layer["type"] != "Add" or len(ins) != 2:
                continue
            for which in (0, 1):
                src = name_of.get(id(ins[which]))
                if src not in by_name or by_name[src] in drop:
                    continue
                ck = by_name[src]
                cplan, cins, cout, clayer = self.steps[ck]
                if clayer["type"] != "Conv2D" or len(consumers.get(src, [])) != 1 or src in outputs or cout.shape != out.shape:
                    continue
                try:
                    fused = capi.chain_plan(self.ctx, [cplan, plan])
                except capi.SnnHipError as e:
                    if e.code != capi.E_UNSUPPORTED:
                        raise
                    continue
                if fused.num_steps() != 1 or "+add" not in fused.describe():
                    fused.destroy()
                    continue
                replace[k] = (fused, [cins[0], ins[1 - which]], out, dict(layer, type="Conv2D", fused="conv+add"))
                drop.add(ck)
                self.fused_adds.append(layer["name"])
                break
        self.steps = [replace.get(k, st) for k, st in enumerate(self.steps) if k not in drop]

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
