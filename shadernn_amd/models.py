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


def to_json_dict(net, width, height):
    """The reference's JSON model: Layer_0 is the InputLayer, layer i consumes layer i-1 (all nets here are chains)."""
    layers = net["layers"]
    out = {"numLayers": {"count": len(layers) + 1}, "inputRange": "[0,1]",
           "Layer_0": {"name": "input_1", "type": "InputLayer", "Input Width": int(width), "Input Height": int(height),
                       "outputPlanes": int(net["input_channels"]), "numInputs": 0, "inputId": []}}
    for i, l in enumerate(layers, start=1):
        o = {"name": l["name"], "numInputs": 1, "inputId": [i - 1], "inputPlanes": int(l["ic"]), "outputPlanes": int(l["oc"])}
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
        else:
            o["type"] = t
            for k, v in l.items():
                if k not in ("type", "name", "ic", "oc") and not isinstance(v, np.ndarray):
                    o[k] = v
        out["Layer_%d" % i] = o
    return out


def write_json(net, width, height, path):
    with open(path, "w") as f:
        json.dump(to_json_dict(net, width, height), f)
    return path
