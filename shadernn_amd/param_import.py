"""ncnn `.param` topology -> the graph-net dict of models.py (and from there the reference's JSON model format).

The reference's model zoo ships its graphs as ncnn text `.param` files (modelzoo/*/*.param) next to Git-LFS weight blobs; its
converter (tools/convertTool) turns such graphs into SNN JSON with BatchNorm / ReLU / Clip folded into the producing Conv2D and the
ReLU after an element-wise add folded into the Add layer (cf. resnet18Test.cpp:84-140).  This module does the same folding on the
topology and attaches synthetic weights (the LFS blobs are not available), so the zoo's ResNet-18 / MobileNetV2 / Candy graphs run
through the HIP backend exactly as the reference would lay them out.

.param grammar (ncnn): line 1 magic 7767517, line 2 "<layers> <blobs>", then per layer
    <type> <name> <n_in> <n_out> <in blobs...> <out blobs...> <id>=<value> ...
"""
import numpy as np

from . import models

MAGIC = "7767517"


def parse_param(text):
    """-> list of {"type", "name", "inputs", "outputs", "params": {int: float|int}} in file order."""
    lines = [l.strip() for l in text.splitlines() if l.strip()]
    if not lines or lines[0] != MAGIC:
        raise ValueError("not an ncnn .param file (magic %r)" % (lines[0] if lines else ""))
    ops = []
    for line in lines[2:]:
        tok = line.split()
        t, name, nin, nout = tok[0], tok[1], int(tok[2]), int(tok[3])
        blobs = tok[4:4 + nin + nout]
        params = {}
        for kv in tok[4 + nin + nout:]:
            k, v = kv.split("=", 1)
            if k.startswith("-"):  # array parameter: -23300=n,v0,v1,... (not used by the zoo graphs handled here)
                continue
            params[int(k)] = float(v) if ("." in v or "e" in v.lower()) else int(v)
        ops.append({"type": t, "name": name, "inputs": blobs[:nin], "outputs": blobs[nin:], "params": params})
    return ops


def _same_or_valid(pad, k):
    if pad == -233 or (k > 1 and pad == k // 2) or k == 1:
        return "same"
    if pad == 0:
        return "valid"
    raise ValueError("explicit convolution padding %d for kernel %d is not one of the reference's 'same' / 'valid' cases" % (pad, k))


def from_ops(ops, name="imported", seed=1, input_shape=None):
    """ops (parse_param output or the tests/golden fixture) -> models.py graph net with synthetic weights.
    input_shape = (H, W, C) overrides / supplies the Input layer's dims (Candy's .param has none)."""
    rng = np.random.default_rng(seed)
    consumers = {}
    for op in ops:
        for b in op["inputs"]:
            consumers.setdefault(b, []).append(op)
    producer_of = {}   # blob -> layer name in the emitted net ("input" for the model input)
    channels = {}      # blob -> channel count
    spatial1 = set()   # blobs known to be 1x1 (after global pooling)
    layers = []
    skip = set()
    in_hw = None
    in_c = None

    def sole_consumer(blob, types):
        c = consumers.get(blob, [])
        return c[0] if len(c) == 1 and c[0]["type"] in types and id(c[0]) not in skip else None

    def fold_activation(blob):
        """Follows BatchNorm -> Clip(-inf, 6) -> ReLU chains hanging off `blob`; returns (bn?, activation, alpha, last blob)."""
        bn, act, alpha = False, "linear", 0.0
        nxt = sole_consumer(blob, ("BatchNorm",))
        if nxt:
            bn = True
            skip.add(id(nxt))
            blob = nxt["outputs"][0]
        nxt = sole_consumer(blob, ("Clip",))
        if nxt and nxt["params"].get(1, 0) == 6.0:
            act = "relu6"
            skip.add(id(nxt))
            blob = nxt["outputs"][0]
        nxt = sole_consumer(blob, ("ReLU",))
        if nxt:
            slope = nxt["params"].get(0, 0.0)
            if act == "linear":
                act, alpha = ("leakyRelu", float(slope)) if slope else ("relu", 0.0)
            skip.add(id(nxt))
            blob = nxt["outputs"][0]
        for t, a in (("Sigmoid", "sigmoid"), ("TanH", "tanh")):
            nxt = sole_consumer(blob, (t,))
            if nxt and act == "linear":
                act = a
                skip.add(id(nxt))
                blob = nxt["outputs"][0]
        return bn, act, alpha, blob

    def emit(layer, ins, out_blob, c):
        layer["inputs"] = [producer_of[b] for b in ins]
        layers.append(layer)
        producer_of[out_blob] = layer["name"]
        channels[out_blob] = c

    for op in ops:
        if id(op) in skip:
            continue
        t, p, nm = op["type"], op["params"], op["name"]
        ins, outs = op["inputs"], op["outputs"]
        if t == "Input":
            w, h, c = p.get(0, 0), p.get(1, 0), p.get(2, 0)
            if input_shape:
                h, w, c = input_shape
            if not (w and h and c):
                raise ValueError("the Input layer carries no dims: pass input_shape=(H, W, C)")
            in_hw, in_c = (h, w), c
            producer_of[outs[0]], channels[outs[0]] = "input", c
        elif t == "Split":
            for o in outs:
                producer_of[o], channels[o] = producer_of[ins[0]], channels[ins[0]]
                if ins[0] in spatial1:
                    spatial1.add(o)
        elif t in ("Convolution", "ConvolutionDepthWise"):
            oc, k, s = p[0], p.get(1, 1), p.get(3, 1)
            ic = channels[ins[0]]
            if p.get(2, 1) != 1:
                raise ValueError("%s: dilation %d is not supported by the reference's Conv2D" % (nm, p[2]))
            if t == "Convolution" and k == 1 and ins[0] in spatial1:
                # 1x1 convolution on the globally pooled 1x1 map == a dense layer (MobileNetV2's classifier); Softmax folds into it
                act = "linear"
                nxt = sole_consumer(outs[0], ("Softmax",))
                blob = outs[0]
                if nxt:
                    act = "softmax"
                    skip.add(id(nxt))
                    blob = nxt["outputs"][0]
                emit(models._op("Flatten", nm + "_flatten", ic), ins, outs[0] + "#flat", ic)
                d = models._dense(rng, nm, ic, oc, act)
                emit(d, [outs[0] + "#flat"], blob, oc)
                continue
            bn, act, alpha, blob = fold_activation(outs[0])
            fused = p.get(9, 0)  # ncnn's fused activation_type: 1 relu, 3 clip, 4 sigmoid (2 = leaky relu carries its slope in an array parameter)
            if fused and act == "linear":
                if fused not in (1, 4):
                    raise ValueError("%s: fused activation_type %d" % (nm, fused))
                act = {1: "relu", 4: "sigmoid"}[fused]
            pad = _same_or_valid(p.get(4, 0), k)
            if t == "ConvolutionDepthWise":
                if p.get(7, 1) != ic or oc != ic:
                    raise ValueError("%s: grouped convolution other than depthwise" % nm)
                layer = models._depthwise(rng, nm, ic, k, act, stride=s, bn=bn, padding=pad)
            else:
                if p.get(6, ic * oc * k * k) != ic * oc * k * k:
                    raise ValueError("%s: weight size %d != %d*%d*%d*%d" % (nm, p[6], oc, ic, k, k))
                layer = models._conv(rng, nm, ic, oc, k, act, stride=s, bn=bn, padding=pad, bias=bool(p.get(5, 0)) or bn)
            if act == "leakyRelu":
                layer["alpha"] = alpha
            emit(layer, ins, blob, oc)
        elif t == "BatchNorm":  # not preceded by a convolution
            c = channels[ins[0]]
            _, act, alpha, blob = fold_activation(outs[0])
            emit(models._op("BatchNormalization", nm, c, bn=models._bn(rng, c), activation=act, alpha=alpha), ins, blob, c)
        elif t in ("ReLU", "Clip", "Sigmoid", "TanH"):  # stand-alone activation
            c = channels[ins[0]]
            act = {"ReLU": "leakyRelu" if p.get(0, 0.0) else "relu", "Clip": "relu6", "Sigmoid": "sigmoid", "TanH": "tanh"}[t]
            emit(models._op("Activation", nm, c, activation=act, alpha=float(p.get(0, 0.0)) if t == "ReLU" else 0.0), ins, outs[0], c)
        elif t == "BinaryOp":
            if p.get(0, 0) != 0 or len(ins) != 2:
                raise ValueError("%s: only the two-input add BinaryOp exists in the reference (AddLayer)" % nm)
            c = channels[ins[0]]
            _, act, alpha, blob = fold_activation(outs[0])
            emit(models._op("Add", nm, c, activation=act, alpha=alpha), ins, blob, c)
        elif t == "Pooling":
            c = channels[ins[0]]
            if p.get(4, 0):  # global pooling
                if p.get(0, 0) != 1:
                    raise ValueError("%s: global max pooling" % nm)
                emit(models._op("AdaptiveAvgPool2d", nm, c, pool=1), ins, outs[0], c)
                spatial1.add(outs[0])
            else:
                kind = "MaxPooling2D" if p.get(0, 0) == 0 else "AveragePooling2D"
                pad_mode = p.get(5, 0)
                emit(models._op(kind, nm, c, pool=p.get(1, 1), stride=p.get(2, 1), padding="same" if pad_mode in (2, 3) else "valid"), ins, outs[0], c)
                if ins[0] in spatial1:
                    spatial1.add(outs[0])
        elif t in ("Reshape", "Flatten"):
            c = channels[ins[0]]
            if consumers.get(outs[0]):  # feeds a dense layer: HWC flatten
                emit(models._op("Flatten", nm, c), ins, outs[0], c)
            else:                       # trailing reshape of the result: a view
                producer_of[outs[0]], channels[outs[0]] = producer_of[ins[0]], c
        elif t == "InnerProduct":
            oc = p[0]
            inu = p[2] // oc
            act = "linear"
            blob = outs[0]
            nxt = sole_consumer(blob, ("Softmax",))
            if nxt:
                act = "softmax"
                skip.add(id(nxt))
                blob = nxt["outputs"][0]
            emit(models._dense(rng, nm, inu, oc, act), ins, blob, oc)
        elif t == "Padding":
            c = channels[ins[0]]
            mode = {0: "constant", 1: "replicate", 2: "reflect"}[p.get(4, 0)]
            emit(models._op("Pad", nm, c, padding=[[p.get(0, 0), p.get(1, 0)], [p.get(2, 0), p.get(3, 0)]], mode=mode), ins, outs[0], c)
        elif t == "InstanceNorm":
            c = channels[ins[0]]
            _, act, alpha, blob = fold_activation(outs[0])
            emit(models._op("InstanceNorm", nm, c, beta=rng.uniform(-0.2, 0.2, c).astype(np.float32), gamma=rng.uniform(0.5, 1.5, c).astype(np.float32),
                            epsilon=float(p.get(1, 1e-5)), activation=act, alpha=alpha), ins, blob, c)
        elif t == "Interp":
            c = channels[ins[0]]
            if p.get(1, 1.0) != p.get(2, 1.0):
                raise ValueError("%s: anisotropic resize" % nm)
            emit(models._op("UpSampling2D", nm, c, scaleFactor=float(p.get(1, 2.0)), interpolation="nearest" if p.get(0, 1) == 1 else "bilinear"), ins, outs[0], c)
        elif t == "Concat":
            if p.get(0, 0) != 0 or len(ins) != 2:
                raise ValueError("%s: only the two-input channel concat exists in the reference (ConcatenateLayer, concatenation.h:33-37)" % nm)
            c0, c1 = channels[ins[0]], channels[ins[1]]
            emit(models._op("Concatenate", nm, c0 + c1, c0=c0, c1=c1), ins, outs[0], c0 + c1)
            layers[-1]["ic"] = c0
        elif t == "Softmax":
            raise ValueError("%s: a Softmax that does not follow a dense layer has no layer in the reference" % nm)
        else:
            raise ValueError("%s: ncnn layer type %s has no counterpart in the HIP backend (reference layers: Conv2D, DepthwiseConv2D, Dense, Add, "
                             "pooling, Flatten, Pad, InstanceNorm, UpSampling2D, BatchNormalization, Activation, Concatenate)" % (nm, t))
    if in_c is None:
        raise ValueError("no Input layer")
    net = {"name": name, "input_channels": in_c, "input_hw": in_hw, "layers": layers}
    consumed = {n for l in layers for n in l["inputs"]}
    leaves = [l["name"] for l in layers if l["name"] not in consumed]
    if len(leaves) > 1:  # YOLOv3-tiny: two detection heads (the reference joins them in its CPU YOLO layer, yololayer.cpp:177-226)
        net["outputs"] = leaves
    return net


def import_param(path, seed=1, input_shape=None):
    with open(path) as f:
        return from_ops(parse_param(f.read()), name=path.rsplit("/", 1)[-1].rsplit(".", 1)[0], seed=seed, input_shape=input_shape)
