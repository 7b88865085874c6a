"""Synthetic model definitions (random-init weights of the reference's architectures) and a writer for the reference's
.json static-graph format.

The reference's model zoo consists of Git-LFS pointers, so every model here is synthesised deterministically.
JSON schema: reference core/src/ic2/modelparser.cpp:39-44 (numLayers), :480-497 (InputLayer), :574-781 (Conv2D),
:783-985 (Depthwise), :499-572 (Dense); ESPCN topology: reference demo/modelInferenceESPCN.py:49-71.
"""
import json

import numpy as np


def _conv(rng, name, ic, oc, k, act, stride=1, bn=False, padding="same", bias=True):
    fan_in = ic * k * k
    w = (rng.standard_normal((oc, ic, k, k)) / np.sqrt(fan_in)).astype(np.float32)
    layer = {"type": "Conv2D", "name": name, "w": w, "b": rng.uniform(-0.1, 0.1, oc).astype(np.float32) if bias else None, "kernel": k,
             "stride": stride, "padding": padding, "activation": act, "bn": None, "ic": ic, "oc": oc}
    if bn:
        layer["bn"] = _bn(rng, oc)
    return layer


def _bn(rng, c):
    return {"beta": rng.uniform(-0.1, 0.1, c).astype(np.float32), "gamma": rng.uniform(0.5, 1.5, c).astype(np.float32),
            "mean": rng.uniform(-0.1, 0.1, c).astype(np.float32), "var": rng.uniform(0.5, 1.5, c).astype(np.float32)}


def _depthwise(rng, name, c, k, act, stride=1, bn=False, padding="same"):
    w = (rng.standard_normal((c, k, k)) / np.sqrt(k * k)).astype(np.float32)
    return {"type": "DepthwiseConv2D", "name": name, "w": w, "b": rng.uniform(-0.1, 0.1, c).astype(np.float32), "kernel": k, "stride": stride,
            "padding": padding, "activation": act, "bn": _bn(rng, c) if bn else None, "ic": c, "oc": c}


def _dense(rng, name, inu, outu, act):
    w = (rng.standard_normal((outu, inu)) / np.sqrt(inu)).astype(np.float32)  # flat kernel is read as [Out][In]
    return {"type": "Dense", "name": name, "w": w, "b": rng.uniform(-0.1, 0.1, outu).astype(np.float32), "units": outu, "activation": act,
            "ic": inu, "oc": outu}


def espcn_weights(seed=1):
    """ESPCN 2x: conv5x5 1->16 relu, conv3x3 16->16 relu, conv3x3 16->4 linear, depth-to-space(2)+tanh."""
    rng = np.random.default_rng(seed)
    return {"name": "ESPCN_2X", "input_channels": 1,
            "layers": [_conv(rng, "conv2d", 1, 16, 5, "relu"), _conv(rng, "conv2d_1", 16, 16, 3, "relu"), _conv(rng, "conv2d_2", 16, 4, 3, "linear"),
                       {"type": "Subpixel", "name": "subpixel", "ic": 4, "oc": 1}]}


def single_conv(seed=1, ic=3, oc=64, k=3, act="relu", stride=1, bn=False):
    """BASELINE config 1: one 3x3 Conv2D 3->64."""
    rng = np.random.default_rng(seed)
    return {"name": "single_conv", "input_channels": ic, "layers": [_conv(rng, "conv2d", ic, oc, k, act, stride=stride, bn=bn)]}


# ---- graph-shaped nets: a layer may name its producers in "inputs" (default: the previous layer; "input" = the model input) ----

def _op(t, name, c, inputs=None, **kw):
    d = {"type": t, "name": name, "ic": c, "oc": c}
    if inputs is not None:
        d["inputs"] = list(inputs)
    d.update(kw)
    return d


def resnet18(seed=1, num_classes=1000, width=64, in_channels=3):
    """ResNet-18 as the reference's converter emits it (BN + ReLU fused into Conv2D, ReLU fused into Add; resnet18Test.cpp:84-140):
    conv7x7/2 -> maxpool3/2 -> 4 stages x 2 basic blocks (1x1/2 downsample at the stage entry) -> global average pool -> flatten -> dense.
    `width` scales the channel counts (64 = the real model)."""
    rng = np.random.default_rng(seed)
    L = [_conv(rng, "conv1", in_channels, width, 7, "relu", stride=2, bn=True), _op("MaxPooling2D", "pool1", width, pool=3, stride=2, padding="same")]
    prev, c = "pool1", width
    for stage in range(4):
        oc = width << stage
        for blk in range(2):
            stride = 2 if (stage > 0 and blk == 0) else 1
            tag = "l%d_b%d" % (stage + 1, blk)
            a = _conv(rng, tag + "_conv1", c, oc, 3, "relu", stride=stride, bn=True)
            a["inputs"] = [prev]
            b = _conv(rng, tag + "_conv2", oc, oc, 3, "linear", bn=True)
            L += [a, b]
            skip = prev
            if stride != 1 or c != oc:
                d = _conv(rng, tag + "_down", c, oc, 1, "linear", stride=stride, bn=True)
                d["inputs"] = [prev]
                L.append(d)
                skip = d["name"]
            L.append(_op("Add", tag + "_add", oc, inputs=[b["name"], skip], activation="relu"))
            prev, c = tag + "_add", oc
    L += [_op("AdaptiveAvgPool2d", "avgpool", c, pool=1), _op("Flatten", "flatten", c), _dense(rng, "fc", c, num_classes, "softmax")]
    return {"name": "resnet18", "input_channels": in_channels, "layers": L}


def mobilenetv2(seed=1, num_classes=1000, width_mult=1.0, in_channels=3):
    """MobileNetV2 (relu6, BN fused): conv3x3/2 -> 17 inverted-residual blocks (expand 1x1, depthwise 3x3, project 1x1, Add when
    stride 1 and equal widths) -> conv1x1 1280 -> global average pool -> flatten -> dense."""
    rng = np.random.default_rng(seed)
    ch = lambda c: max(8, int(c * width_mult + 4) // 8 * 8)
    c = ch(32)
    L = [_conv(rng, "conv_stem", in_channels, c, 3, "relu6", stride=2, bn=True)]
    prev = "conv_stem"
    idx = 0
    for t, oc_, n, s in [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]:
        oc = ch(oc_)
        for i in range(n):
            stride = s if i == 0 else 1
            tag = "b%02d" % idx
            idx += 1
            x = prev
            hid = c * t
            if t != 1:
                e = _conv(rng, tag + "_expand", c, hid, 1, "relu6", bn=True)
                e["inputs"] = [x]
                L.append(e)
                x = e["name"]
            dw = _depthwise(rng, tag + "_dw", hid, 3, "relu6", stride=stride, bn=True)
            dw["inputs"] = [x]
            pj = _conv(rng, tag + "_project", hid, oc, 1, "linear", bn=True)
            L += [dw, pj]
            out = pj["name"]
            if stride == 1 and c == oc:
                L.append(_op("Add", tag + "_add", oc, inputs=[out, prev], activation="linear"))
                out = tag + "_add"
            prev, c = out, oc
    last = ch(1280) if width_mult > 1.0 else max(8, int(1280 * min(width_mult, 1.0)) // 8 * 8) if width_mult < 1.0 else 1280
    hd = _conv(rng, "conv_head", c, last, 1, "relu6", bn=True)
    hd["inputs"] = [prev]
    L += [hd, _op("AdaptiveAvgPool2d", "avgpool", last, pool=1), _op("Flatten", "flatten", last), _dense(rng, "fc", last, num_classes, "softmax")]
    return {"name": "mobilenetv2", "input_channels": in_channels, "layers": L}


def style_net(seed=1, width=16, in_channels=3):
    """A small Candy-shaped style-transfer net (fast-neural-style): reflect pads + convs + instance norm, one residual block,
    nearest upsampling -- every operator of BASELINE config 5 at toy size (the real candy-9 graph has 16 convs / 15 instance norms)."""
    rng = np.random.default_rng(seed)
    inorm = lambda name, c, act: _op("InstanceNorm", name, c, beta=rng.uniform(-0.2, 0.2, c).astype(np.float32), gamma=rng.uniform(0.5, 1.5, c).astype(np.float32),
                                     epsilon=1e-5, activation=act)
    pad = lambda name, c, p: _op("Pad", name, c, padding=[[p, p], [p, p]], mode="reflect")
    w = width
    L = [pad("pad1", in_channels, 4), _conv(rng, "conv1", in_channels, w, 9, "linear", padding="valid"), inorm("in1", w, "relu"),
         pad("pad2", w, 1), _conv(rng, "conv2", w, 2 * w, 3, "linear", stride=2, padding="valid"), inorm("in2", 2 * w, "relu"),
         # residual block: zero-padded "same" convs -- under the reference's size rule a Pad layer grows the tensor and a "valid"
         # conv does not shrink it back, so pad+valid inside a skip connection would not line up with the skip
         _conv(rng, "r_conv1", 2 * w, 2 * w, 3, "linear"), inorm("r_in1", 2 * w, "relu"),
         _conv(rng, "r_conv2", 2 * w, 2 * w, 3, "linear"), inorm("r_in2", 2 * w, "linear"),
         _op("Add", "r_add", 2 * w, inputs=["r_in2", "in2"], activation="linear"),
         _op("UpSampling2D", "up1", 2 * w, scaleFactor=2.0, interpolation="nearest"),
         pad("pad3", 2 * w, 1), _conv(rng, "conv3", 2 * w, w, 3, "linear", padding="valid"), inorm("in3", w, "relu"),
         pad("pad4", w, 4), _conv(rng, "conv4", w, in_channels, 9, "linear", padding="valid")]
    return {"name": "style_net", "input_channels": in_channels, "layers": L}


def _deconv(rng, name, ic, oc, k, act, stride=2, bn=False, padding="same"):
    """Conv2DTranspose (Conv2DTransposeDesc : Conv2DDesc, deconv2d.h:21): same JSON fields as Conv2D, weights flat OIHW."""
    layer = _conv(rng, name, ic, oc, k, act, stride=stride, bn=bn, padding=padding)
    layer["type"] = "Conv2DTranspose"
    return layer


def unet(seed=1, base=64, in_channels=1, depth=4):
    """The zoo's U-Net shape (modelzoo/U-Net/unet.param): two 3x3 relu convs per level, 2x2 max-pool down, nearest 2x up + 2x2... the zoo
    graph uses UpSampling2D + 3x3 conv, then Concatenate with the encoder skip; 1x1 sigmoid head."""
    rng = np.random.default_rng(seed)
    L, skips, c, prev = [], [], in_channels, "input"
    for d in range(depth):
        w = base << d
        L += [_conv(rng, "enc%d_a" % d, c, w, 3, "relu"), _conv(rng, "enc%d_b" % d, w, w, 3, "relu")]
        skips.append(("enc%d_b" % d, w))
        L.append(_op("MaxPooling2D", "pool%d" % d, w, pool=2, stride=2, padding="same"))
        c = w
    w = base << depth
    L += [_conv(rng, "mid_a", c, w, 3, "relu"), _conv(rng, "mid_b", w, w, 3, "relu")]
    c = w
    for d in reversed(range(depth)):
        sk, w = skips[d]
        L += [_op("UpSampling2D", "up%d" % d, c, scaleFactor=2.0, interpolation="nearest"), _conv(rng, "up%d_conv" % d, c, w, 3, "relu"),
              _op("Concatenate", "cat%d" % d, 2 * w, inputs=["up%d_conv" % d, sk], c0=w, c1=w)]
        L[-1]["ic"] = w
        L += [_conv(rng, "dec%d_a" % d, 2 * w, w, 3, "relu"), _conv(rng, "dec%d_b" % d, w, w, 3, "relu")]
        c = w
    L.append(_conv(rng, "head", c, 1, 1, "sigmoid"))
    return {"name": "unet", "input_channels": in_channels, "layers": L}


def output_names(net):
    """The net's result layers: net["outputs"] when it has several heads, else the last layer."""
    return list(net.get("outputs", [net["layers"][-1]["name"]]))


def producers(net):
    """[(layer, [producer names])] with the chain default made explicit."""
    out, prev = [], "input"
    for l in net["layers"]:
        out.append((l, list(l.get("inputs", [prev]))))
        prev = l["name"]
    return out


def to_json_dict(net, width, height):
    """The reference's JSON model: Layer_0 is the InputLayer; "inputId" lists the producer layers (modelparser.cpp:133-143)."""
    layers = net["layers"]
    out = {"numLayers": {"count": len(layers) + 1}, "inputRange": "[0,1]",
           "Layer_0": {"name": "input_1", "type": "InputLayer", "Input Width": int(width), "Input Height": int(height),
                       "outputPlanes": int(net["input_channels"]), "numInputs": 0, "inputId": []}}
    ids = {"input": 0}
    for i, (l, ins) in enumerate(producers(net), start=1):
        ids[l["name"]] = i
        o = {"name": l["name"], "numInputs": len(ins), "inputId": [ids[n] for n in ins], "inputPlanes": int(l["ic"]), "outputPlanes": int(l["oc"])}
        t = l["type"]
        if t in ("Conv2D", "DepthwiseConv2D"):
            o["type"] = t
            o.update({"activation": l["activation"], "padding": l["padding"], "kernel_size": int(l["kernel"]), "strides": int(l["stride"]),
                      "useBias": "True" if l["b"] is not None else "False", "useBatchNormalization": "True" if l["bn"] else "False"})
            if t == "Conv2D":
                kernel = l["w"].reshape(-1)  # flat OIHW (modelparser.cpp:639-657)
            else:
                kernel = np.transpose(l["w"], (1, 2, 0)).reshape(-1)  # flat HWC (modelparser.cpp:842-850)
            o["weights"] = {"kernel": [float(v) for v in kernel], "bias": [float(v) for v in (l["b"] if l["b"] is not None else [])]}
            if l["bn"]:
                o["batchNormalization"] = {"beta": [float(v) for v in l["bn"]["beta"]], "gamma": [float(v) for v in l["bn"]["gamma"]],
                                           "moving_mean": [float(v) for v in l["bn"]["mean"]],
                                           "moving_variance": [float(v) for v in l["bn"]["var"]]}
            if l["activation"] == "leakyRelu":
                o["leakyReluAlpha"] = float(l.get("alpha", 0.1))
        elif t == "Dense":
            o["type"] = "Dense"
            o.update({"units": int(l["units"]), "activation": l["activation"], "useBias": "True",
                      "weights": {"kernel": [float(v) for v in l["w"].reshape(-1)], "bias": [float(v) for v in l["b"]]}})
        elif t == "Subpixel":
            o["type"] = "Lambda"  # dispatched by NAME (modelparser.cpp:82-84, layerFactory.cpp:147-149)
            o["name"] = "subpixel"
        elif t in ("MaxPooling2D", "AveragePooling2D", "AdaptiveAvgPool2d"):
            o.update({"type": t, "pool": [int(l["pool"]), int(l["pool"])]})
            if t != "AdaptiveAvgPool2d":
                o.update({"stride": int(l["stride"]), "padding": l["padding"]})
        elif t in ("Add", "Activation", "Flatten"):
            o.update({"type": t, "activation": l.get("activation", "linear")})
            if l.get("activation") == "leakyRelu":
                o["leakyReluAlpha"] = float(l.get("alpha", 0.1))
        elif t == "BatchNormalization":
            o.update({"type": t, "activation": l.get("activation", "linear"),
                      "batchNormalization": {"beta": [float(v) for v in l["bn"]["beta"]], "gamma": [float(v) for v in l["bn"]["gamma"]],
                                             "moving_mean": [float(v) for v in l["bn"]["mean"]], "moving_variance": [float(v) for v in l["bn"]["var"]]}})
        elif t == "Pad":
            o.update({"type": "Pad", "padding": l["padding"], "mode": l["mode"]})
        elif t == "InstanceNorm":
            o.update({"type": "InstanceNormalization", "epsilon": float(l["epsilon"]), "activation": l["activation"],
                      "weights": {"bias": [float(v) for v in l["beta"]], "scale": [float(v) for v in l["gamma"]]}})
        elif t == "UpSampling2D":
            o.update({"type": t, "scaleFactor": float(l["scaleFactor"]), "interpolation": l["interpolation"]})
        elif t == "Concatenate":
            o.update({"type": t, "inputPlanes": int(l["c0"])})
        elif t == "Conv2DTranspose":
            o.update({"type": t, "activation": l["activation"], "padding": l["padding"], "kernel_size": int(l["kernel"]), "strides": int(l["stride"]),
                      "useBias": "True" if l["b"] is not None else "False", "useBatchNormalization": "True" if l["bn"] else "False",
                      "weights": {"kernel": [float(v) for v in l["w"].reshape(-1)], "bias": [float(v) for v in (l["b"] if l["b"] is not None else [])]}})
            if l["bn"]:
                o["batchNormalization"] = {"beta": [float(v) for v in l["bn"]["beta"]], "gamma": [float(v) for v in l["bn"]["gamma"]],
                                           "moving_mean": [float(v) for v in l["bn"]["mean"]], "moving_variance": [float(v) for v in l["bn"]["var"]]}
            if l["activation"] == "leakyRelu":
                o["leakyReluAlpha"] = float(l.get("alpha", 0.1))
        else:
            o["type"] = t
            for k, v in l.items():
                if k not in ("type", "name", "ic", "oc") and not isinstance(v, np.ndarray):
                    o[k] = v
        out["Layer_%d" % i] = o
    return out


def write_json(net, width, height, path, bin_weights=False):
    """Writes the reference's JSON model.  bin_weights=True: convolution / depthwise / transposed-convolution / dense parameters go to a
    "<name>.bin" side file of raw float32 named by numLayers.bin_file_name (modelparser.cpp:234-257) instead of inline JSON arrays --
    read back sequentially in layer order: kernel (Conv2D flat OIHW :617-658, depthwise CHW :826-840, dense [In][Out] rows :527-535),
    bias when useBias, then BN as gamma, beta, mean, variance (:685-757).  What real-size models (ResNet-18: 11.7 M weights) need."""
    d = to_json_dict(net, width, height)
    if bin_weights:
        bin_path = path[:-5] + ".bin" if path.endswith(".json") else path + ".bin"
        import os

        d["numLayers"]["bin_file_name"] = os.path.basename(bin_path)
        with open(bin_path, "wb") as fb:
            for i, l in enumerate(net["layers"], start=1):
                o = d["Layer_%d" % i]
                t = l["type"]
                if t not in ("Conv2D", "DepthwiseConv2D", "Conv2DTranspose", "Dense"):
                    continue
                o.pop("weights", None)
                o.pop("batchNormalization", None)
                np.ascontiguousarray(l["w"], dtype=np.float32).tofile(fb)  # Conv2D: OIHW; depthwise: CHW; dense: the flat kernel
                if l.get("b") is not None:
                    np.ascontiguousarray(l["b"], dtype=np.float32).tofile(fb)
                if l.get("bn"):
                    for k in ("gamma", "beta", "mean", "var"):
                        np.ascontiguousarray(l["bn"][k], dtype=np.float32).tofile(fb)
    with open(path, "w") as f:
        json.dump(d, f)
    return path


def zoo(name, input_shape, seed=1):
    """A graph of the reference's model zoo (modelzoo/*.param, topologies parsed once into shadernn_amd/data/zoo_topologies.json by
    tests/golden/make_zoo_topologies.py) with synthetic weights: e.g. zoo("candy-9_simplified-opt", (720, 1280, 3)) = BASELINE configs[4]."""
    import json
    import os

    from . import param_import

    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "zoo_topologies.json")))[name]
    ops = [{"type": o["type"], "name": o["name"], "inputs": o["inputs"], "outputs": o["outputs"], "params": {int(k): v for k, v in o["params"].items()}}
           for o in fx["ops"]]
    return param_import.from_ops(ops, name=name, seed=seed, input_shape=input_shape)
