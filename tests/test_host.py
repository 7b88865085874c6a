"""The C++ host mirror of the reference API (shadernn_amd/host -> libsnn_core.so) driven the way the reference's harnesses drive
the original: JSON model -> ModelParser -> layer DAG -> InferenceGraph -> MixedInferenceCore::create/run -> .dump files."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = dict(rtol=1e-4, atol=1e-4)


def _json(tmp_path, net, w, h):
    from shadernn_amd import models

    return models.write_json(net, w, h, str(tmp_path / (net["name"] + ".json")))


def test_core_library_exports_declared_symbols(built):
    from shadernn_amd import host

    l = host.lib()
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "snn_c.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(snn_[a-z0-9_]+)\s*\(", txt)))
    assert declared == sorted(host.SIGNATURES)
    for name in declared:
        assert hasattr(ctypes.CDLL(host.LIB_PATH), name)
    assert l is not None


def test_json_number_grammar_overflow_and_underflow(built):
    """The loader's number parser (std::from_chars reports overflow AND underflow as out-of-range): overflow saturates, underflow goes to +-0 / a
    denormal -- what the reference's picojson (strtod) yields -- and never to +-inf."""
    import math

    from shadernn_amd import host

    assert host.json_number("1.5e3") == 1500.0 and host.json_number("-0.25") == -0.25 and host.json_number("0") == 0.0
    assert host.json_number("1e400") == math.inf and host.json_number("-1e400") == -math.inf
    assert host.json_number("1e-400") == 0.0 and host.json_number("-1e-400") == 0.0 and math.copysign(1.0, host.json_number("-1e-400")) == -1.0
    assert host.json_number("1.5e400") == math.inf and host.json_number("1.5e-400") == 0.0  # the out-of-range fallback parses the '.' in the "C" locale (strtod_l)
    assert host.json_number("4.9e-324") == 5e-324 and 0.0 < host.json_number("2e-310") < 1e-300  # denormals survive
    for bad in ("1.", ".5", "+1", "01", "1e", "nan", "0x10", ""):
        assert host.json_number(bad) is None, bad


def test_parser_and_graph_on_cpu(built, tmp_path):
    """No GPU needed: plan creation is deferred to the backend, like pipeline creation in the reference."""
    from shadernn_amd import host, models

    net = models.espcn_weights(seed=1)
    rows = host.graph_summary(_json(tmp_path, net, 96, 72), 96, 72, 1)
    assert [r["name"].split("] ")[1] for r in rows] == ["InputLayer", "Conv2D", "Conv2D", "Conv2D", "subpixel"]  # 5 layers
    assert rows[0]["name"].endswith("layer [00] InputLayer")
    assert [r["dims"] for r in rows] == [(96, 72, 1), (96, 72, 16), (96, 72, 16), (96, 72, 4), (192, 144, 1)]
    assert [r["inputs"] for r in rows] == [[-1], [0], [1], [2], [3]]
    assert all(r["loc"] == 4 for r in rows)  # LayerExecutionType::GPU_HIP


def test_parser_shape_rules_on_cpu(built, tmp_path):
    from shadernn_amd import host, models

    rng = np.random.default_rng(0)
    net = {"name": "shapes", "input_channels": 3, "layers": [
        models._conv(rng, "c7", 3, 8, 7, "relu", stride=2, bn=True),      # 224 -> 112 (float rule)
        models._depthwise(rng, "dw", 8, 3, "relu6", stride=2, bn=True),   # 112 -> 56
        models._conv(rng, "pw", 8, 12, 1, "linear", stride=2),            # 56 -> 28, 1x1 has no padding
        models._conv(rng, "k4", 12, 4, 4, "leakyRelu"),                   # even kernel: asymmetric padding, same size
        models._conv(rng, "valid", 4, 4, 3, "relu", padding="valid"),     # Q20: "valid" keeps the size
    ]}
    net["layers"][3]["alpha"] = 0.2
    rows = host.graph_summary(_json(tmp_path, net, 224, 224), 224, 224, 3)
    assert [r["dims"] for r in rows] == [(224, 224, 3), (112, 112, 8), (56, 56, 8), (28, 28, 12), (28, 28, 4), (28, 28, 4)]


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [False, True])
def test_espcn_json_model_end_to_end(ctx, tmp_path, fuse):
    from shadernn_amd import host, models

    net = models.espcn_weights(seed=1)
    W, H = 96, 72
    x = np.random.default_rng(7767517).random((1, H, W, 1), dtype=np.float32)
    m = host.Model(_json(tmp_path, net, W, H), W, H, 1, fuse_chains=fuse, profiling=True)
    y = m(x)
    want, layers = O.forward(net, x, return_layers=True)
    np.testing.assert_allclose(y, want[0], **TOL)
    st = m.stages()
    assert len(st) == 5
    if fuse:
        assert [s["fused_away"] for s in st] == [False, True, True, True, False]  # the fused plan sits at the last stage of its group
    else:
        for i in range(1, 5):  # layer-by-layer, like resnet18Test.cpp:84-140
            np.testing.assert_allclose(m.stage_output(i), layers[i - 1][0], **TOL)
    stats = m.time_stats()
    assert any("Conv2D" in k for k in stats) and all(v >= 0 for v in stats.values())
    y2 = m(x)  # run() is repeatable
    np.testing.assert_array_equal(y, y2)
    m.close()


@pytest.mark.gpu
def test_layer_dumps_match_reference_format(ctx, tmp_path, monkeypatch):
    """dumpOutputs: '<dir>/<layer name> pass[0].dump', 32-byte header + RGBA32F [D][H][W][4] (image.cpp:216-245)."""
    from shadernn_amd import host, models

    monkeypatch.setenv("SNN_OUTPUT_DIR", str(tmp_path / "dump"))
    net = models.espcn_weights(seed=1)
    W, H = 40, 24
    x = np.random.default_rng(3).random((1, H, W, 1), dtype=np.float32)
    path = _json(tmp_path, net, W, H)
    m = host.Model(path, W, H, 1, dump_outputs=True)
    m(x)
    _, layers = O.forward(net, x, return_layers=True)
    names = ["[01] Conv2D", "[02] Conv2D", "[03] Conv2D", "[04] subpixel"]
    for name, exp in zip(names, layers):
        f = str(tmp_path / "dump" / ("ESPCN_2X.json layer %s pass[0].dump" % name))
        raw = open(f, "rb").read()
        w, h, d, c, px = host.read_dump(f)
        assert raw[:32].rstrip(b"\0").decode() == "%d %d %d %d" % (w, h, d, c) and len(raw) == 32 + w * h * d * 16
        assert (h, w) == exp.shape[1:3] and d == (exp.shape[3] + 3) // 4 and c == 4 * d
        np.testing.assert_allclose(host.c4hw4_to_nhwc(px, exp.shape[3]), exp[0], **TOL)
        assert os.path.exists(f.replace("pass[0].dump", "pass[0]_input.dump"))
    m.close()


@pytest.mark.gpu
def test_conv_test_with_layer_reference_defaults(ctx, tmp_path, monkeypatch):
    """convolutionTest.cpp main(): 8x8x128 -> 1, k=1, all-ones input, weights from SRAND(7767517), bias 0, identity BN."""
    from shadernn_amd import host

    monkeypatch.setenv("SNN_OUTPUT_DIR", str(tmp_path))
    os.makedirs(str(tmp_path), exist_ok=True)
    w = O.reference_rand(7767517, 128).reshape(1, 128, 1, 1)
    x = np.ones((8, 8, 128), np.float32)
    bn = {"gamma": np.ones(1, np.float32), "mean": np.zeros(1, np.float32), "var": np.ones(1, np.float32), "beta": np.zeros(1, np.float32)}
    f = host.conv_test_with_layer(x, w, np.zeros(1, np.float32), stride=1, pad=0, bn=bn)
    assert f.endswith("resnet18_cifar10_0223.json layer [01] Conv2D pass[0].dump")
    _, _, d, c, px = host.read_dump(f)
    assert (d, c) == (1, 4)
    want = O.conv2d(x[None], w, np.zeros(1, np.float32), 1, (0, 0, 0, 0), "constant", "", 0.0,
                    {"beta": bn["beta"], "gamma": bn["gamma"], "mean": bn["mean"], "var": bn["var"]})
    np.testing.assert_allclose(host.c4hw4_to_nhwc(px, 1), want[0], **TOL)


@pytest.mark.gpu
@pytest.mark.parametrize("k,stride,pad", [(3, 1, 0), (3, 2, 1), (5, 1, 2), (3, 1, 2)])
def test_conv_test_with_layer_variants(ctx, tmp_path, monkeypatch, k, stride, pad):
    from shadernn_amd import host

    monkeypatch.setenv("SNN_OUTPUT_DIR", str(tmp_path))
    rng = np.random.default_rng(k * 10 + stride)
    x = rng.standard_normal((9, 11, 5)).astype(np.float32)
    w = (rng.standard_normal((7, 5, k, k)) * 0.2).astype(np.float32)
    b = (rng.standard_normal(7) * 0.1).astype(np.float32)
    f = host.conv_test_with_layer(x, w, b, stride=stride, pad=pad)
    _, _, _, _, px = host.read_dump(f)
    p = k // 2
    want = O.conv2d(x[None], w, b, stride, (p, p, p, p), ["constant", "replicate", "reflect"][pad], "")
    np.testing.assert_allclose(host.c4hw4_to_nhwc(px, 7), want[0], **TOL)


@pytest.mark.gpu
def test_mixed_model_with_depthwise_and_dense(ctx, tmp_path):
    """A MobileNet-style fragment through the JSON loader: conv(bn,relu6) -> depthwise(bn,relu6) -> 1x1 conv -> dense(softmax)."""
    from shadernn_amd import host, models

    rng = np.random.default_rng(11)
    net = {"name": "frag", "input_channels": 3, "layers": [
        models._conv(rng, "stem", 3, 8, 3, "relu6", stride=2, bn=True),
        models._depthwise(rng, "dw", 8, 3, "relu6", stride=1, bn=True),
        models._conv(rng, "pw", 8, 12, 1, "linear", bn=True),
        models._dense(rng, "fc", 8 * 8 * 12, 10, "softmax"),
    ]}
    x = rng.random((1, 16, 16, 3), dtype=np.float32)
    m = host.Model(_json(tmp_path, net, 16, 16), 16, 16, 3)
    y = m(x)
    want = O.forward(net, x)
    np.testing.assert_allclose(y.reshape(-1), want.reshape(-1), **TOL)
    assert abs(float(y.sum()) - 1.0) < 1e-5
    m.close()


# ---- graph-shaped models with the element-wise / pooling / shape operators ---------------------------------------

def _small_nets():
    from shadernn_amd import models

    return [(models.resnet18(seed=2, num_classes=10, width=8), 64, 64), (models.mobilenetv2(seed=3, num_classes=10, width_mult=0.25), 64, 64),
            (models.style_net(seed=4, width=8), 32, 24)]


def test_graph_models_parse_on_cpu(built, tmp_path):
    """ResNet-18 / MobileNetV2 / style-net topologies through ModelParser + the graph builder: layer types, DAG edges, shape rules."""
    from shadernn_amd import host, models

    net, w, h = _small_nets()[0]
    rows = host.graph_summary(_json(tmp_path, net, w, h), w, h, 3)
    names = [r["name"].split("] ")[1] for r in rows]
    assert names[:3] == ["InputLayer", "Conv2D", "MaxPooling2D"] and names[-3:] == ["AdaptiveAvgPool2d", "Flatten", "Dense"]
    assert names.count("Add") == 8 and names.count("Conv2D") == 20
    assert rows[1]["dims"] == (32, 32, 8) and rows[2]["dims"] == (16, 16, 8)       # conv7x7/2, maxpool3/2 "same" (float rule)
    assert rows[-3]["dims"] == (1, 1, 64) and rows[-2]["dims"][0] == 64 and rows[-1]["dims"][0] == 10
    adds = [r for r in rows if "Add" in r["name"]]
    assert all(len(r["inputs"]) == 2 for r in adds)
    net, w, h = _small_nets()[2]
    rows = host.graph_summary(_json(tmp_path, net, w, h), w, h, 3)
    dims = {r["name"].split("] ")[1] + str(i): r["dims"] for i, r in enumerate(rows)}
    assert rows[1]["dims"] == (40, 32, 3)        # reflect pad 4
    assert rows[2]["dims"] == (40, 32, 8)        # Q20: the "valid" 9x9 conv keeps the padded size
    assert any("UpSampling2D" in k for k in dims) and any("InstanceNorm" in k for k in dims)


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1, 2], ids=["resnet18", "mobilenetv2", "style_net"])
def test_graph_models_json_end_to_end(ctx, tmp_path, which):
    """JSON -> C++ host mirror -> HIP plans (batch 1), every stage output against the oracle's (layer-by-layer, resnet18Test.cpp:84-140 style)."""
    from shadernn_amd import host

    net, w, h = _small_nets()[which]
    x = np.random.default_rng(5).random((1, h, w, 3), dtype=np.float32)
    m = host.Model(_json(tmp_path, net, w, h), w, h, 3, fuse_chains=False)
    y = m(x)
    want, named = O.forward(net, x, return_named=True)
    st = m.stages()
    assert len(st) == len(net["layers"]) + 1
    seen = set()
    for i in range(1, len(st)):  # stages are in topological order (dp.cpp:389-429); the layer id is in the stage name
        lid = int(re.search(r"layer \[(\d+)\]", st[i]["name"]).group(1))
        seen.add(lid)
        exp = named[net["layers"][lid - 1]["name"]]
        np.testing.assert_allclose(m.stage_output(i).reshape(-1), exp.reshape(-1), err_msg="stage %d %s" % (i, st[i]["name"]), **TOL)
    assert seen == set(range(1, len(net["layers"]) + 1))
    np.testing.assert_allclose(y.reshape(-1), want.reshape(-1), **TOL)
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1, 2], ids=["resnet18", "mobilenetv2", "style_net"])
def test_graph_runner_batched(ctx, which):
    """The same graphs at batch 3 through per-layer C-ABI plans (what tools/bench_models.py times at batch 32)."""
    import shadernn_amd as snn

    net, w, h = _small_nets()[which]
    x = np.random.default_rng(6).random((3, h, w, 3), dtype=np.float32)
    r = snn.GraphRunner(ctx, net, 3, h, w)
    y = r(x)
    want = O.forward(net, x)
    np.testing.assert_allclose(y.reshape(3, -1), want.reshape(3, -1), **TOL)


def test_bin_side_file_parses_like_inline_weights(built, tmp_path):
    """models.write_json(bin_weights=True): the ".bin" side file named by numLayers.bin_file_name (modelparser.cpp:234-257) gives the same graph
    as inline JSON arrays (weight VALUES are compared on the GPU, test_host_batched_graphs)."""
    from shadernn_amd import host, models

    for which in (0, 1):
        net, w, h = _small_nets()[which]
        a = host.graph_summary(models.write_json(net, w, h, str(tmp_path / "inline.json")), w, h, 3)
        b = host.graph_summary(models.write_json(net, w, h, str(tmp_path / "side.json"), bin_weights=True), w, h, 3)
        assert os.path.getsize(str(tmp_path / "side.bin")) > 1000 and os.path.getsize(str(tmp_path / "side.json")) < os.path.getsize(str(tmp_path / "inline.json")) // 4
        strip = lambda rows: [(r["name"].split("] ")[1], r["dims"], r["inputs"]) for r in rows]
        assert strip(a) == strip(b)


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("which", [0, 1, 2], ids=["resnet18", "mobilenetv2", "style_net"])
def test_host_batched_graphs(ctx, tmp_path, which, fuse):
    """The C++ host twin of test_graph_runner_batched: snn_model_create4(batch=3) -- every stage tensor carries the batch (the 4th texture
    dimension the reference fixes to 1, core.cpp:371) -- from a JSON + .bin model, layer by layer against the oracle; with fuse the stage DAG
    goes through snnhip_graph_fuse (residual conv+add pairs, pad/upsample strings)."""
    from shadernn_amd import host, models

    net, w, h = _small_nets()[which]
    if which == 2:  # 32+ channels: the convolutions run on the MFMA kernel, the one with the fused pad / upsampling staging path
        net, w, h = models.style_net(seed=4, width=32), 40, 32
    B = 3
    x = np.random.default_rng(6).random((B, h, w, 3), dtype=np.float32)
    path = models.write_json(net, w, h, str(tmp_path / (net["name"] + "_b.json")), bin_weights=True)
    m = host.Model(path, w, h, 3, fuse_chains=fuse, batch=B)
    y = m(x)
    want, named = O.forward(net, x, return_named=True)
    assert y.shape[0] == B
    np.testing.assert_allclose(y.reshape(B, -1), want.reshape(B, -1), **TOL)
    st = m.stages()
    folded = [s for s in st if s["fused_away"]]
    if fuse:
        assert folded, m.describe()  # ResNet / MobileNetV2: the residual convolutions; style net: pads and the upsampling
        assert any("+add" in d or "+pad" in d or "chain{" in d for _, _, d, _, _ in m.plan_steps()), m.plan_steps()
    else:
        assert not folded
    for i in range(1, len(st)):
        if st[i]["fused_away"]:
            continue
        lid = int(re.search(r"layer \[(\d+)\]", st[i]["name"]).group(1))
        exp = named[net["layers"][lid - 1]["name"]]
        got = m.stage_output(i)
        assert got is not None and got.shape[0] == B
        np.testing.assert_allclose(got.reshape(B, -1), exp.reshape(B, -1), err_msg=st[i]["name"], **TOL)
    f, b = m.cost()
    assert f > 0 and b > 0
    m.close()


@pytest.mark.gpu
def test_host_batched_capture_and_profile(ctx, tmp_path):
    """Batch 4 with capture_graph (one hipGraph launch per inference) equals the per-launch run; the per-launch profiling hooks bench.py uses
    report every kernel of the fused graph."""
    from shadernn_amd import host, models

    net, w, h = _small_nets()[0]
    path = models.write_json(net, w, h, str(tmp_path / "r18.json"), bin_weights=True)
    x = np.random.default_rng(8).random((4, h, w, 3), dtype=np.float32)
    ref = host.Model(path, w, h, 3, batch=4)
    cap = host.Model(path, w, h, 3, batch=4, capture_graph=True)
    y0 = ref(x)
    for _ in range(3):
        np.testing.assert_array_equal(cap(x), y0)
    steps = ref.plan_steps()
    assert len(steps) >= 20 and all(f >= 0 and b > 0 for _, _, _, f, b in steps)
    ref.profile(True)
    ref.run()
    ref.run()
    ref.profile(False)
    for stage, step, desc, _, _ in steps:
        ms, n = ref.profile_read(stage, step)
        assert n == 2 and ms > 0, desc
    for mm in (ref, cap):
        mm.close()


# ---- Concatenate / Conv2DTranspose / Unary / YOLO through the host mirror, pre- and post-processing ----

def _deconv_concat_net(seed=11):
    from shadernn_amd import models

    rng = np.random.default_rng(seed)
    L = [models._conv(rng, "down", 4, 8, 3, "relu", stride=2),
         models._deconv(rng, "up", 8, 8, 4, "relu", stride=2, bn=True),
         models._conv(rng, "skip", 4, 12, 3, "leakyRelu"),
         models._op("Concatenate", "cat", 20, inputs=["up", "skip"], c0=8, c1=12),
         models._op("Unary", "copy", 20),
         models._conv(rng, "head", 20, 4, 1, "tanh")]
    L[2]["inputs"] = ["input"]
    L[2]["alpha"] = 0.1
    L[3]["ic"] = 8
    return {"name": "deconv_concat", "input_channels": 4, "layers": L}


def _yolo_net(seed=12):
    """416 x 416 x 3 -> strided convs -> a 26 x 26 and a 13 x 13 head of 3 x 6 channels -> the CPU YOLO layer."""
    from shadernn_amd import models

    rng = np.random.default_rng(seed)
    L, c = [], 3
    for i, oc in enumerate([8, 8, 16, 16]):
        L.append(models._conv(rng, "c%d" % i, c, oc, 3, "leakyRelu", stride=2))
        L[-1]["alpha"] = 0.1
        c = oc
    fine = models._conv(rng, "fine", c, 18, 1, "linear")
    coarse_f = models._conv(rng, "c4", c, 32, 3, "leakyRelu", stride=2)
    coarse_f["inputs"], coarse_f["alpha"] = ["c3"], 0.1
    coarse = models._conv(rng, "coarse", 32, 18, 1, "linear")
    for hd in (fine, coarse):
        hd["w"] *= 20.0 / 6.0
        hd["b"] = rng.uniform(-1.0, 1.0, 18).astype(np.float32)
    L += [fine, coarse_f, coarse, models._op("YOLO", "yolo", 18, inputs=["coarse", "fine"])]
    return {"name": "yolo_mini", "input_channels": 3, "layers": L}


def test_rank4_graphs_parse_on_cpu(built, tmp_path):
    from shadernn_amd import host

    net = _deconv_concat_net()
    rows = host.graph_summary(_json(tmp_path, net, 16, 12), 16, 12, 4)
    names = [r["name"].split("] ")[1] for r in rows]
    assert names == ["InputLayer", "Conv2D", "Conv2D", "Conv2DTranspose", "Concatenate", "Unary", "Conv2D"] or names[1:4].count("Conv2DTranspose") == 1
    by = {r["name"].split("] ")[1] + ":" + str(r["dims"]): r for r in rows}
    assert any(k.startswith("Conv2DTranspose:(16, 12, 8)") for k in by)   # stride 2 "same": 2 x the 8 x 6 input (deconv2dGL.cpp:345-355)
    cat = [r for r in rows if "Concatenate" in r["name"]][0]
    assert cat["dims"] == (16, 12, 20) and len(cat["inputs"]) == 2
    rows = host.graph_summary(_json(tmp_path, _yolo_net(), 416, 416), 416, 416, 3)
    yolo = rows[-1]
    assert "YOLO" in yolo["name"] and yolo["dims"][:2] == (600, 1) and len(yolo["inputs"]) == 2   # yololayer.h:44-48: 100 boxes x 6
    assert yolo["loc"] != rows[-2]["loc"]                                                          # the one CPU stage


def test_yolo_decode_host_vs_numpy_restatement(built):
    from shadernn_amd import host

    rng = np.random.default_rng(3)
    heads = [rng.normal(-2.0, 2.5, (1, 13, 13, 18)).astype(np.float32), rng.normal(-2.0, 2.5, (1, 26, 26, 18)).astype(np.float32)]
    want = np.array(O.yolo_decode(heads), np.float32).reshape(-1, 6)
    got = host.yolo_decode(heads[0], heads[1], 416, max_rows=4096)
    assert len(want) > 20 and got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_deconv_concat_unary_json_end_to_end(ctx, tmp_path):
    from shadernn_amd import host

    net = _deconv_concat_net()
    x = np.random.default_rng(5).random((1, 12, 16, 4), dtype=np.float32)
    m = host.Model(_json(tmp_path, net, 16, 12), 16, 12, 4, fuse_chains=False)
    y = m(x)
    want, named = O.forward(net, x, return_named=True)
    st = m.stages()
    for i in range(1, len(st)):
        lid = int(re.search(r"layer \[(\d+)\]", st[i]["name"]).group(1))
        exp = named[net["layers"][lid - 1]["name"]]
        np.testing.assert_allclose(m.stage_output(i).reshape(-1), exp.reshape(-1), err_msg=st[i]["name"], **TOL)
    np.testing.assert_allclose(y.reshape(-1), want.reshape(-1), **TOL)
    m.close()


@pytest.mark.gpu
def test_classifier_output_is_one_based_argmax(ctx, tmp_path):
    from shadernn_amd import host

    net, w, h = _small_nets()[0]
    x = np.random.default_rng(7).random((1, h, w, 3), dtype=np.float32)
    m = host.Model(_json(tmp_path, net, w, h), w, h, 3)
    m.set_type("classification")
    y = m(x)
    assert m.classifier_output() == int(np.argmax(y.reshape(-1))) + 1 == O.argmax(O.forward(net, x)) + 1
    m.close()


@pytest.mark.gpu
def test_yolo_model_detections(ctx, tmp_path):
    from shadernn_amd import host

    net = _yolo_net()
    x = np.random.default_rng(8).random((1, 416, 416, 3), dtype=np.float32)
    m = host.Model(_json(tmp_path, net, 416, 416), 416, 416, 3)
    m.set_type("detection")
    m.upload(x)
    m.run()
    body = dict(net, layers=net["layers"][:-1])
    _, named = O.forward(body, x, threads=8, return_named=True)
    want = np.array(O.yolo_decode([named["coarse"], named["fine"]]), np.float32).reshape(-1, 6)
    got = m.detections(max_rows=8192)
    # the device heads agree with the oracle's to ~1e-4, so a box right at the confidence / IoU threshold may flip: match boxes by geometry
    assert len(want) > 100 and abs(len(want) - len(got)) <= 0.01 * len(want)
    d = np.abs(want[:, None, 2:6] - got[None, :, 2:6]).max(axis=2)
    j = d.argmin(axis=1)
    ok = (d[np.arange(len(want)), j] < 1e-3) & (np.abs(want[:, 1] - got[j, 1]) < 1e-3) & (want[:, 0] == got[j, 0])
    assert ok.mean() >= 0.99, ok.mean()
    assert np.all(np.diff(got[:, 1]) <= 1e-7)  # sorted by score, like Nms() leaves them
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ch", [1, 3, 4])
def test_u8_image_preprocessing_on_device(ctx, tmp_path, ch):
    """8-bit image -> convertToRGBA32FAndNormalize -> ImageTexture::resize(1/255 norms) -> model, vs the oracle's restatements."""
    from shadernn_amd import host, models

    net = models.single_conv(seed=9, ic=4, oc=8, k=3, act="relu")
    img = np.random.default_rng(ch).integers(0, 256, (45, 70, ch), dtype=np.uint8)
    m = host.Model(_json(tmp_path, net, 32, 24), 32, 24, 4)
    means, norms = (10.0, 20.0, 30.0, 0.0), (1.0, 1.0, 1.0, 1.0)
    rn = (1 / 255.0,) * 4
    m.upload_u8(img, means, norms, (0, 0, 0, 0), rn)
    m.run()
    t = O.resize(O.image_u8(img[None], means, norms), 24, 32, (0, 0, 0, 0), rn, True)
    np.testing.assert_allclose(m.output().reshape(-1), O.forward(net, t).reshape(-1), rtol=1e-4, atol=1e-4)
    m.close()


@pytest.mark.gpu
def test_graph_capture_with_u8_uploads_between_runs(ctx, tmp_path):
    """capture_graph=True while snn_model_upload_input_u8 re-creates the input tensor before every run (loadU8AndNormalize + resize free and
    re-allocate it): a recorded launch sequence is replayed only when the input's DEVICE buffer, extent and type are the recorded ones, so every
    run must see the image just uploaded -- images of changing source size force changing intermediate allocations in between."""
    from shadernn_amd import host, models

    net = {"name": "two_convs", "input_channels": 4,
           "layers": models.single_conv(seed=9, ic=4, oc=8, k=3, act="relu")["layers"] + models.single_conv(seed=10, ic=8, oc=4, k=3, act="tanh")["layers"]}
    net["layers"][1]["name"] = "conv2d_b"
    path = _json(tmp_path, net, 32, 24)
    m = host.Model(path, 32, 24, 4, capture_graph=True)
    ref = host.Model(path, 32, 24, 4)
    rn = (1 / 255.0,) * 4
    rng = np.random.default_rng(77)
    for k, (ih, iw) in enumerate([(45, 70), (45, 70), (24, 32), (60, 33), (45, 70), (24, 32), (24, 32)]):
        img = rng.integers(0, 256, (ih, iw, 4), dtype=np.uint8)
        for mm in (m, ref):
            mm.upload_u8(img, (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0), rn)
            mm.run()
        t = O.resize(O.image_u8(img[None], (0, 0, 0, 0), (1, 1, 1, 1)), 24, 32, (0, 0, 0, 0), rn, True)
        np.testing.assert_array_equal(m.output(), ref.output(), err_msg="run %d" % k)
        np.testing.assert_allclose(m.output().reshape(-1), O.forward(net, t).reshape(-1), rtol=1e-4, atol=1e-4, err_msg="run %d" % k)
        # plain float uploads in between go to the same texture too
        x = rng.random((1, 24, 32, 4), dtype=np.float32)
        np.testing.assert_array_equal(m(x), ref(x))
    m.close()
    ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize("half", [False, True])
def test_style_net_with_chain_fusion_through_host(ctx, tmp_path, half):
    """fuse_chains=True: HipBackend::finalizeStages hands the linear runs to snnhip_chain_plan_create, whose rule D folds every reflect Pad
    into the convolution behind it; the result must not change."""
    from shadernn_amd import host, models

    net, w, h = models.style_net(seed=4, width=32), 40, 32  # 32+ channels: the convolutions run on the MFMA kernel, which has the fused-pad path
    x = np.random.default_rng(15).random((1, h, w, 3), dtype=np.float32)
    m = host.Model(_json(tmp_path, net, w, h), w, h, 3, fuse_chains=True, prefer_half=half)
    y = m(x)
    assert "+pad(reflect)" in m.describe() or any(s["fused_away"] for s in m.stages()), m.describe()
    if half:
        want = O.forward(net, x, fp16=True)
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(y.reshape(-1) / scale, want.reshape(-1) / scale, rtol=3e-2, atol=3e-2)
    else:
        np.testing.assert_allclose(y.reshape(-1), O.forward(net, x).reshape(-1), **TOL)
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1, 2], ids=["resnet18", "mobilenetv2", "style_net"])
def test_host_graph_capture_replays_with_new_inputs(ctx, tmp_path, which):
    """capture_graph: the first run() records the launch sequence as a hipGraph, later runs replay it -- new input data (same texture) must flow through,
    and a replaced input texture (upload_u8 path) must trigger a re-recording."""
    import time

    from shadernn_amd import host

    net, w, h = _small_nets()[which]
    path = _json(tmp_path, net, w, h)
    m = host.Model(path, w, h, 3, capture_graph=True)
    ref = host.Model(path, w, h, 3)
    for seed in (21, 22, 23):
        x = np.random.default_rng(seed).random((1, h, w, 3), dtype=np.float32)
        y = m(x)
        np.testing.assert_allclose(y.reshape(-1), O.forward(net, x).reshape(-1), **TOL)
        np.testing.assert_array_equal(y, ref(x))
    # host-side cost of one run(): replay = one launch call instead of one per layer
    for mm in (m, ref):
        mm.run()
    t = []
    for mm in (m, ref):
        t0 = time.perf_counter()
        for _ in range(20):
            mm.run()
        t.append((time.perf_counter() - t0) / 20)
    print("run(): graph replay %.1f us, per-layer launches %.1f us" % (t[0] * 1e6, t[1] * 1e6))
    m.close()
    ref.close()


@pytest.mark.gpu
def test_calculate_layer_json_end_to_end(ctx, tmp_path):
    """The moonwellbox "Calculate" layer (fs_calculation.glsl: rgb / illumination) through JSON -> host mirror -> HIP, and through GraphRunner."""
    import shadernn_amd as snn
    from shadernn_amd import host, models

    rng = np.random.default_rng(31)
    conv = models._conv(rng, "feat", 3, 12, 3, "sigmoid")  # strictly positive: the divisor is channel 8
    calc = models._op("Calculate", "calc", 12)
    calc["oc"] = 4
    net = {"name": "calc_net", "input_channels": 3, "layers": [conv, calc]}
    x = rng.random((1, 20, 28, 3), dtype=np.float32)
    want = O.forward(net, x)
    assert want.shape == (1, 20, 28, 4) and np.all(want[..., 3] == 0)
    m = host.Model(_json(tmp_path, net, 28, 20), 28, 20, 3)
    np.testing.assert_allclose(m(x).reshape(-1), want.reshape(-1), rtol=1e-4, atol=1e-4)
    m.close()
    r = snn.GraphRunner(ctx, net, 1, 20, 28)
    np.testing.assert_allclose(r(x).reshape(-1), want.reshape(-1), rtol=1e-4, atol=1e-4)


def test_pool_and_cli_fail_loudly_without_a_gpu(built, tmp_path):
    """snn_pool_create / lib/snn_run on a box without a GPU: an error code, never a silent CPU path; the CLI rejects a malformed command line (rc 2)."""
    import subprocess

    import torch

    from shadernn_amd import host, models

    cli = os.path.join(ROOT, "shadernn_amd", "lib", "snn_run")
    assert os.path.exists(cli), "build_host() did not produce lib/snn_run"
    r = subprocess.run([cli, "--w", "8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 2 and "usage: snn_run" in r.stderr
    if torch.cuda.is_available():
        return
    path = models.write_json(models.espcn_weights(seed=1), 16, 12, str(tmp_path / "m.json"), bin_weights=True)
    try:
        host.Pool(path, 16, 12, 1, devices=[0, 0], global_batch=4)
    except (RuntimeError, SystemExit):
        pass
    else:
        raise AssertionError("snn_pool_create succeeded without a GPU")


def test_stage_graphs_of_the_bench_configs_match_the_stored_ones():
    """dp::loadFromJsonModel + dp::generateInferenceGraph (host/dp.cpp; reference core/src/ic2/dp.cpp:115-167, 389-640): stage order, stage names,
    execution types, output dims and input references of the five BASELINE configs against tests/golden/graph_stages.json, which was written with the
    round-4 implementation (tests/golden/make_graph_stages.py) -- the stage order names dump files and timers, so a rewrite of the file must not move it."""
    import json
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_graph_stages

    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graph_stages.json")))
    got = make_graph_stages.stage_graphs()
    assert sorted(got) == sorted(want)
    for cfg in want:
        assert len(got[cfg]) == len(want[cfg]), cfg
        for g, w in zip(got[cfg], want[cfg]):
            assert g == w, (cfg, g, w)


def test_malformed_layer_graphs_stop_with_a_message_that_names_the_layer(built, tmp_path):
    """A model whose layers form a cycle, or that names an inbound layer it does not have, must end in SNN_RIP with a readable reason -- the reference's
    own error style (a FATAL log line, then abort), not an exception escaping through the C ABI (round-5 review: stageOf.at() threw std::out_of_range).
    Each case in a child process: the abort is the expected outcome."""
    import json
    import subprocess
    import sys

    from shadernn_amd import models

    good = models.write_json(models.espcn_weights(seed=1), 16, 12, str(tmp_path / "good.json"), bin_weights=True)
    cases = {}
    d = json.load(open(good))
    d["Layer_1"]["inputId"] = [2]          # conv2d reads conv2d_1, which reads conv2d: a cycle
    cases["cycle"] = (d, "the layer graph has a cycle", "[01] Conv2D")
    d = json.load(open(good))
    d["Layer_3"]["inputId"] = [7]          # no such layer
    cases["dangling"] = (d, "names inbound layer 7", "layers 0..4")
    child = ("import sys; sys.path.insert(0, %r); from shadernn_amd import host; print(len(host.graph_summary(sys.argv[1], 16, 12, 1)))" % ROOT)
    r = subprocess.run([sys.executable, "-c", child, good], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "5", r.stderr[-1000:]
    for name, (model, reason, detail) in cases.items():
        path = str(tmp_path / (name + ".json"))
        model["numLayers"]["bin_file_name"] = "good.bin"  # (the weights of the untouched model, beside it)
        json.dump(model, open(path, "w"))
        r = subprocess.run([sys.executable, "-c", child, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        assert r.returncode != 0, name
        assert r.returncode < 0 or r.returncode == 134, (name, r.returncode)   # SIGABRT: SNN_RIP
        text = r.stderr + r.stdout
        assert reason in text and detail in text, (name, text[-1500:])
        assert "out_of_range" not in text and "terminate called" not in text, text[-1500:]
