"""ncnn .param importer: the reference's model-zoo graphs (fixture shadernn_amd/data/zoo_topologies.json, generated from
modelzoo/*.param by tests/golden/make_zoo_topologies.py) -> graph nets with the converter's folding rules -> oracle / HIP."""
import json
import os
from collections import Counter

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = dict(rtol=1e-4, atol=1e-4)


def _zoo(name, input_shape=None, seed=1):
    from shadernn_amd import param_import

    fx = json.load(open(os.path.join(ROOT, "shadernn_amd", "data", "zoo_topologies.json")))[name]
    ops = [{"type": o["type"], "name": o["name"], "inputs": o["inputs"], "outputs": o["outputs"], "params": {int(k): v for k, v in o["params"].items()}}
           for o in fx["ops"]]
    return param_import.from_ops(ops, name=name, seed=seed, input_shape=input_shape)


def test_parse_param_text():
    from shadernn_amd import param_import

    txt = "7767517\n3 3\nInput in 0 1 a 0=8 1=6 2=3\nConvolution c 1 1 a b 0=4 1=3 3=2 4=1 5=1 6=108\nReLU r 1 1 b c 0=1.000000e-01\n"
    ops = param_import.parse_param(txt)
    assert [o["type"] for o in ops] == ["Input", "Convolution", "ReLU"] and ops[1]["params"] == {0: 4, 1: 3, 3: 2, 4: 1, 5: 1, 6: 108}
    net = param_import.from_ops(ops)
    assert net["input_hw"] == (6, 8) and len(net["layers"]) == 1
    l = net["layers"][0]
    assert (l["type"], l["ic"], l["oc"], l["kernel"], l["stride"], l["padding"], l["activation"]) == ("Conv2D", 3, 4, 3, 2, "same", "leakyRelu")
    assert abs(l["alpha"] - 0.1) < 1e-7
    with pytest.raises(ValueError):
        param_import.parse_param("123\n")


def test_zoo_graphs_fold_like_the_reference_converter():
    net = _zoo("resnet18_cifar10")
    c = Counter(l["type"] for l in net["layers"])
    assert net["input_hw"] == (32, 32) and net["input_channels"] == 3
    assert c == Counter({"Conv2D": 20, "Add": 8, "MaxPooling2D": 1, "AveragePooling2D": 1, "Flatten": 1, "Dense": 1})  # BN/ReLU folded away
    assert all(l["activation"] == "relu" for l in net["layers"] if l["type"] == "Add")
    assert net["layers"][0]["kernel"] == 7 and net["layers"][0]["stride"] == 2 and net["layers"][0]["bn"] is not None
    assert net["layers"][-1]["units"] == 10 and net["layers"][-1]["activation"] == "softmax" and net["layers"][-1]["ic"] == 512
    net = _zoo("mobilenetV2")
    c = Counter(l["type"] for l in net["layers"])
    assert c["DepthwiseConv2D"] == 17 and c["Conv2D"] == 36 and c["Add"] == 10 and c["Flatten"] == 1 and c["AdaptiveAvgPool2d"] == 1 and c["Dense"] == 1
    assert {l["activation"] for l in net["layers"] if l["type"] == "DepthwiseConv2D"} == {"relu6"}
    net = _zoo("candy-9_simplified-opt", input_shape=(64, 64, 3))
    c = Counter(l["type"] for l in net["layers"])
    assert c == Counter({"Conv2D": 16, "Pad": 16, "InstanceNorm": 15, "Add": 5, "UpSampling2D": 2})
    assert {l["mode"] for l in net["layers"] if l["type"] == "Pad"} == {"reflect"}


def test_zoo_graph_json_roundtrip_through_host_parser(built, tmp_path):
    """The imported Candy graph written as SNN JSON parses in the C++ host mirror (no GPU): layer types and the Q20 size rule."""
    from shadernn_amd import host, models

    net = _zoo("candy-9_simplified-opt", input_shape=(32, 32, 3))
    path = models.write_json(net, 32, 32, str(tmp_path / "candy.json"))
    rows = host.graph_summary(path, 32, 32, 3)
    assert len(rows) == len(net["layers"]) + 1
    assert rows[1]["dims"] == (40, 40, 3) and rows[2]["dims"] == (40, 40, 32)  # reflect pad 4; "valid" 9x9 keeps the size (Q20)
    assert rows[-1]["dims"][2] == 3


@pytest.mark.gpu
@pytest.mark.parametrize("name,shape,batch", [("resnet18_cifar10", None, 2), ("mobilenetV2", (96, 96, 3), 1), ("candy-9_simplified-opt", (40, 48, 3), 1)])
def test_zoo_graphs_on_gpu_match_oracle(ctx, name, shape, batch):
    """Real-width zoo topologies (11 M / 2.3 M / 1.7 M synthetic parameters) through per-layer HIP plans vs the CPU oracle."""
    import shadernn_amd as snn

    net = _zoo(name, input_shape=shape)
    h, w = net["input_hw"]
    x = np.random.default_rng(9).random((batch, h, w, net["input_channels"]), dtype=np.float32)
    r = snn.GraphRunner(ctx, net, batch, h, w)
    y = r(x)
    want = O.forward(net, x, threads=8)
    assert y.reshape(batch, -1).shape == want.reshape(batch, -1).shape
    np.testing.assert_allclose(y.reshape(batch, -1), want.reshape(batch, -1), **TOL)  # element-wise, no normalisation by max|want|


def test_unet_and_yolo_graphs_import():
    """the zoo's U-Net and YOLOv3-tiny graphs need Concatenate; YOLOv3-tiny keeps its two detection heads as outputs."""
    from shadernn_amd import models

    net = _zoo("unet")
    c = Counter(l["type"] for l in net["layers"])
    assert net["input_hw"] == (256, 256) and net["input_channels"] == 1
    assert c == Counter({"Conv2D": 24, "MaxPooling2D": 4, "UpSampling2D": 4, "Concatenate": 4})
    assert Counter(l["activation"] for l in net["layers"] if l["type"] == "Conv2D") == Counter({"relu": 23, "sigmoid": 1})  # ncnn's fused activation_type
    cat = [l for l in net["layers"] if l["type"] == "Concatenate"][0]
    assert (cat["c0"], cat["c1"], cat["oc"]) == (512, 512, 1024) and len(cat["inputs"]) == 2
    assert models.output_names(net) == [net["layers"][-1]["name"]]
    net = _zoo("yolov3-tiny", input_shape=(416, 416, 3))
    c = Counter(l["type"] for l in net["layers"])
    assert c == Counter({"Conv2D": 13, "MaxPooling2D": 6, "UpSampling2D": 1, "Concatenate": 1})
    assert models.output_names(net) == ["conv2d_9", "conv2d_12"]
    assert {l["activation"] for l in net["layers"] if l["type"] == "Conv2D"} == {"leakyRelu", "linear"}


@pytest.mark.gpu
@pytest.mark.parametrize("name,shape", [("unet", (64, 48, 1)), ("yolov3-tiny", (96, 96, 3))])
def test_unet_and_yolo_on_gpu_match_oracle(ctx, name, shape):
    import shadernn_amd as snn

    net = _zoo(name, input_shape=shape)
    h, w = net["input_hw"]
    x = np.random.default_rng(10).random((1, h, w, net["input_channels"]), dtype=np.float32)
    r = snn.GraphRunner(ctx, net, 1, h, w)
    r(x)
    _, named = O.forward(net, x, threads=8, return_named=True)
    for nm, y in zip(r.output_names, r.outputs()):
        want = named[nm]
        assert y.shape == want.shape
        np.testing.assert_allclose(y, want, err_msg=nm, **TOL)
    if name == "yolov3-tiny":
        assert [o.shape for o in r.outputs()] == [(1, 3, 3, 255), (1, 6, 6, 255)]
