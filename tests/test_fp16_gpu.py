"""SNNHIP_F16 tensors (the reference's RGBA16F / preferHp path): half storage, fp16-input MFMA convolutions with fp32 accumulation,
fp32 epilogues, round-to-nearest stores.  Checked against the oracle run with the same quantisation points (fp16 weights, every stored
activation rounded to half) and, loosely, against the fp32 oracle (the reference's own fp16 tests use 0.1, resnet18Test.cpp)."""
import numpy as np
import pytest

import oracle_lib as O
from test_ops_gpu import _bn, _rand

pytestmark = pytest.mark.gpu
TOLH = dict(rtol=4e-3, atol=4e-3)   # ~4 half ulps at unit scale: summation order + rounding-boundary flips


def _conv16(ctx, x, w, b, stride, pads, pad_mode, act, bn):
    import shadernn_amd as snn

    n, h, ww, _ = x.shape
    plan = snn.conv2d_plan(ctx, n, h, ww, w, b, stride=stride, pads=pads, pad_mode=pad_mode, act=act, bn=bn, dtype=snn.F16)
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    yt = plan(xt)
    assert yt.dtype == snn.F16
    y, desc = yt.numpy(), plan.describe()
    xt.free()
    yt.free()
    plan.destroy()
    return y, desc


CASES = [(2, 20, 24, 64, 64, 3, 1), (1, 30, 30, 3, 32, 9, 1), (2, 17, 23, 32, 64, 3, 2), (1, 12, 14, 128, 128, 3, 1), (2, 9, 11, 20, 33, 3, 1),
         (1, 16, 20, 32, 3, 9, 1), (3, 7, 7, 256, 96, 1, 1), (1, 8, 8, 10, 16, 5, 1)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_fp16_conv_matches_quantised_oracle(ctx, case):
    N, H, W, IC, OC, k, s = case
    x = _rand((N, H, W, IC), 81)
    w = _rand((OC, IC, k, k), 82, 1.0 / np.sqrt(IC * k * k))
    b = _rand((OC,), 83, 0.1)
    bn = _bn(OC, 84)
    pads = O.padding_offsets("same", k)
    for pad_mode, act in (("constant", "relu"), ("reflect", "tanh")):
        y, desc = _conv16(ctx, x, w, b, s, pads, pad_mode, act, bn)
        assert "f16" in desc, desc
        want = O._h(O.conv2d(O._h(x), O._h(w), b, s, pads, pad_mode, act, 0.0, bn))
        np.testing.assert_allclose(y, want, err_msg=desc, **TOLH)
        full = O.conv2d(x, w, b, s, pads, pad_mode, act, 0.0, bn)
        assert np.abs(y - full).max() < 0.05, desc


STREAM16 = [(2, 28, 28, 16, 96, 1, "relu6", True), (2, 28, 28, 24, 144, 1, "relu6", True), (1, 33, 47, 32, 192, 1, "relu", False),
            (2, 14, 14, 96, 24, 1, "", True), (1, 19, 21, 144, 32, 1, "", True), (1, 9, 9, 8, 16, 1, "leakyRelu", True),
            (1, 10, 10, 40, 576, 1, "SiLU", True), (2, 27, 31, 64, 128, 2, "", True), (1, 12, 12, 200, 64, 1, "tanh", False)]


@pytest.mark.parametrize("case", STREAM16, ids=lambda c: "x".join(map(str, c)))
def test_fp16_pointwise_stream_matches_quantised_oracle(ctx, monkeypatch, case):
    """conv1x1_stream's fp16 variant: 16-channel chunks incl. a half-empty last one (IC % 16 == 8), the wave-local
    LDS transpose of the output tile, stride 2; against the quantised oracle and the general fp16 kernel."""
    N, H, W, IC, OC, s, act, use_bn = case
    x = _rand((N, H, W, IC), 91)
    w = _rand((OC, IC, 1, 1), 92, 1.0 / np.sqrt(IC))
    b = _rand((OC,), 93, 0.1)
    bn = _bn(OC, 94) if use_bn else None
    monkeypatch.setenv("SNNHIP_CONV_1X1", "2")
    y, desc = _conv16(ctx, x, w, b, s, (0, 0, 0, 0), "constant", act, bn)
    assert "stream" in desc and "f16" in desc, desc
    want = O._h(O.conv2d(O._h(x), O._h(w), b, s, (0, 0, 0, 0), "constant", act, 0.0, bn))
    assert y.shape == want.shape, desc
    np.testing.assert_allclose(y, want, err_msg=desc, **TOLH)
    monkeypatch.setenv("SNNHIP_CONV_1X1", "0")
    y2, desc2 = _conv16(ctx, x, w, b, s, (0, 0, 0, 0), "constant", act, bn)
    assert "stream" not in desc2, desc2
    np.testing.assert_allclose(y, y2, err_msg=desc + " vs " + desc2, **TOLH)


def test_fp16_pointwise_stream_fused_add(ctx, monkeypatch):
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_CONV_1X1", "2")
    n, h, w, ic, oc = 2, 28, 28, 144, 24
    x, wt, b = _rand((n, h, w, ic), 61), _rand((oc, ic, 1, 1), 62, 1.0 / np.sqrt(ic)), _rand((oc,), 63, 0.1)
    skip, bn = _rand((n, h, w, oc), 64), _bn(oc, 65)
    conv = snn.conv2d_plan(ctx, n, h, w, wt, b, stride=1, pads=(0, 0, 0, 0), act="", bn=bn, dtype=snn.F16)
    assert "stream" in conv.describe(), conv.describe()
    add = snn.add_plan(ctx, n, h, w, oc, act="relu")
    fused = snn.chain_plan(ctx, [conv, add])
    assert fused.num_steps() == 1 and "stream" in fused.describe() and "+add" in fused.describe(), fused.describe()
    xt, st = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16), snn.Tensor.from_numpy(ctx, skip, dtype=snn.F16)
    y = fused([xt, st]).numpy()
    c = O._h(O.conv2d(O._h(x), O._h(wt), b, 1, (0, 0, 0, 0), "constant", "", 0.0, bn))
    want = O._h(O.add_act(c, O._h(skip), "relu", 0.0))
    np.testing.assert_allclose(y, want, err_msg=fused.describe(), **TOLH)
    np.testing.assert_allclose(y, add([conv(xt), st]).numpy(), **TOLH)


def test_fp16_tensor_roundtrip_and_dtype_checks(ctx):
    import shadernn_amd as snn

    a = _rand((2, 5, 7, 12), 85, 3.0)
    t = snn.Tensor.from_numpy(ctx, a, dtype=snn.F16)
    np.testing.assert_array_equal(t.numpy(), a.astype(np.float16).astype(np.float32))
    p32 = snn.conv2d_plan(ctx, 2, 5, 7, _rand((16, 12, 3, 3), 86, 0.1))
    with pytest.raises(snn.SnnHipError):  # fp32 plan, fp16 tensor
        p32(t)


@pytest.mark.parametrize("which", ["resnet18", "mobilenetv2"])
def test_fp16_classifiers_match_quantised_oracle(ctx, which):
    """Depthwise, pooling, flatten and dense also take half tensors: ResNet-18 / MobileNetV2 end to end in fp16."""
    import shadernn_amd as snn
    from shadernn_amd import models

    net = models.resnet18(seed=2, num_classes=10, width=16) if which == "resnet18" else models.mobilenetv2(seed=3, num_classes=10, width_mult=0.5)
    x = np.random.default_rng(7).random((2, 64, 64, 3), dtype=np.float32)
    r = snn.GraphRunner(ctx, net, 2, 64, 64, dtype=snn.F16)
    y = r(x).reshape(2, -1)
    want = O.forward(net, x, fp16=True).reshape(2, -1)
    np.testing.assert_allclose(y, want, rtol=2e-2, atol=4e-3)        # softmax outputs
    np.testing.assert_allclose(y.sum(axis=1), 1.0, atol=4e-3)
    np.testing.assert_allclose(y, O.forward(net, x).reshape(2, -1), atol=0.05)


@pytest.mark.parametrize("which", ["style", "candy", "resnet_trunk", "candy+normfusion"])
def test_fp16_graphs_match_quantised_oracle(ctx, monkeypatch, which):
    """Whole graphs with half tensors: pad / conv / instance norm / add / upsample (Candy's operator set) and a ResNet trunk; Candy also with
    the opt-in Conv2D -> InstanceNorm fusion (chain rule F)."""
    import shadernn_amd as snn
    from shadernn_amd import models
    from test_param_import import _zoo

    fuse_norms = which.endswith("+normfusion")
    if fuse_norms:
        monkeypatch.setenv("SNNHIP_NORM_FUSION", "1")
        which = "candy"

    if which == "style":
        net, (h, w) = models.style_net(seed=4, width=16), (24, 32)
    elif which == "candy":
        net, (h, w) = _zoo("candy-9_simplified-opt", input_shape=(32, 40, 3)), (32, 40)
    else:
        net = models.resnet18(seed=2, num_classes=10, width=16)
        net["layers"] = net["layers"][:-3]  # up to the last residual Add (pooling / dense stay fp32-only)
        h, w = 64, 64
    x = np.random.default_rng(5).random((2, h, w, 3), dtype=np.float32)
    r = snn.GraphRunner(ctx, net, 2, h, w, dtype=snn.F16)
    assert bool(r.fused_norms) == fuse_norms, r.fused_norms
    y = r(x)
    want = O.forward(net, x, fp16=True, threads=8)
    scale = max(1.0, float(np.abs(want).max()))
    assert y.shape == want.shape
    err = np.abs(y - want) / scale
    # instance norm amplifies single half-ulp flips of its input; the bulk must agree to a few ulps, the tail stays small
    assert np.quantile(err, 0.999) < 6e-3 and err.max() < 6e-2, (float(np.quantile(err, 0.999)), float(err.max()))
    full = O.forward(net, x, threads=8)
    assert np.abs(y - full).max() / scale < 0.1  # the reference's fp16 bound


def test_fp16_json_model_through_host_mirror(ctx, tmp_path):
    """preferHp end to end: JSON -> ModelParser (weights truncated to fp16, Q13) -> RGBA16F stage textures -> fp16 plans."""
    from shadernn_amd import host, models

    net = models.style_net(seed=4, width=16)
    w, h = 32, 24
    path = models.write_json(net, w, h, str(tmp_path / "style16.json"))
    x = np.random.default_rng(5).random((1, h, w, 3), dtype=np.float32)
    m = host.Model(path, w, h, 3, fuse_chains=False, prefer_half=True)
    y = m(x)
    # the parser TRUNCATES weights / bias / norm parameters to half (convertToMediumPrecision); give the oracle the same values
    import copy

    q = copy.deepcopy(net)
    trunc = np.vectorize(O.to_medium_precision, otypes=[np.float32])
    for l in q["layers"]:
        for k in ("w", "b", "beta", "gamma"):
            if l.get(k) is not None:
                l[k] = trunc(np.asarray(l[k], np.float32))
    want = O.forward(q, x, fp16=True)
    scale = max(1.0, float(np.abs(want).max()))
    err = np.abs(y.reshape(-1) - want.reshape(-1)) / scale
    assert np.quantile(err, 0.999) < 6e-3 and err.max() < 6e-2, (float(np.quantile(err, 0.999)), float(err.max()))
    m.close()


@pytest.mark.parametrize("fuse", [False, True])
def test_fp16_espcn_json_model_through_host_mirror(ctx, tmp_path, fuse):
    """The headline ESPCN model with preferHp (the reference's RGBA16F path): every layer incl. the Subpixel lambda takes half tensors
    (the fp32-only fusion rules A/B/C step aside), against the oracle with the same quantisation points."""
    import copy

    import shadernn_amd as snn
    from shadernn_amd import host, models

    net = models.espcn_weights(seed=1)
    w, h = 48, 40
    path = models.write_json(net, w, h, str(tmp_path / "espcn16.json"))
    x = np.random.default_rng(8).random((1, h, w, 1), dtype=np.float32)
    m = host.Model(path, w, h, 1, fuse_chains=fuse, prefer_half=True)
    y = m(x)
    assert y.shape == (2 * h, 2 * w, 1) or y.shape == (1, 2 * h, 2 * w, 1)
    q = copy.deepcopy(net)
    trunc = np.vectorize(O.to_medium_precision, otypes=[np.float32])  # the parser truncates weights / bias to half (Q13)
    for l in q["layers"]:
        for k in ("w", "b"):
            if l.get(k) is not None:
                l[k] = trunc(np.asarray(l[k], np.float32))
    want = O.forward(q, x, fp16=True)
    np.testing.assert_allclose(y.reshape(-1), want.reshape(-1), rtol=4e-3, atol=4e-3)
    np.testing.assert_allclose(y.reshape(-1), O.forward(net, x).reshape(-1), atol=0.02)
    m.close()
    # and the stand-alone Subpixel plan on half tensors
    t = np.random.default_rng(9).standard_normal((2, 5, 7, 4)).astype(np.float32)
    p = snn.subpixel_plan(ctx, 2, 5, 7, 4, 2, 0)
    got = p(snn.Tensor.from_numpy(ctx, t, dtype=snn.F16)).numpy()
    np.testing.assert_allclose(got, O._h(O.subpixel(O._h(t), 2, 0)), rtol=1e-3, atol=1e-3)


# ---- conv2d_rowfold.hip: wide-kernel, channel-thin fp16 output layers with the kernel columns folded into the MFMA's N ----
ROWFOLD = [(2, 40, 70, 32, 3, 9, "constant", ""), (1, 33, 61, 32, 3, 9, "reflect", "tanh"), (2, 20, 130, 16, 4, 7, "replicate", "relu"),
           (1, 17, 19, 32, 6, 5, "constant", "sigmoid"), (3, 9, 9, 16, 1, 9, "constant", "")]


@pytest.mark.parametrize("case", ROWFOLD, ids=lambda c: "x".join(map(str, c[:6])) + "_" + c[6])
def test_fp16_rowfold_conv_matches_quantised_oracle_and_thin_kernel(ctx, monkeypatch, case):
    import shadernn_amd as snn

    N, H, W, IC, OC, k, pad_mode, act = case
    x = _rand((N, H, W, IC), 201)
    w = _rand((OC, IC, k, k), 202, 1.0 / np.sqrt(IC * k * k))
    b = _rand((OC,), 203, 0.1)
    bn = _bn(OC, 204)
    pads = O.padding_offsets("same", k)
    y, desc = _conv16(ctx, x, w, b, 1, pads, pad_mode, act, bn)
    assert "conv2d_rowfold" in desc, desc
    want = O._h(O.conv2d(O._h(x), O._h(w), b, 1, pads, pad_mode, act, 0.0, bn))
    np.testing.assert_allclose(y, want, err_msg=desc, **TOLH)
    monkeypatch.setenv("SNNHIP_CONV", "thin" if OC <= 4 else "mfma")  # the kernels these layers ran on before
    y2, desc2 = _conv16(ctx, x, w, b, 1, pads, pad_mode, act, bn)
    assert "rowfold" not in desc2, desc2
    np.testing.assert_allclose(y, y2, err_msg=desc + " vs " + desc2, rtol=4e-3, atol=4e-3)


MARCH = [(1, 100, 131, 32, 3, 9, "reflect", "", "3"), (2, 61, 200, 32, 3, 9, "constant", "tanh", ""), (1, 75, 57, 16, 4, 7, "replicate", "relu", "2"),
         (1, 9, 64, 32, 2, 5, "constant", "", ""), (2, 130, 113, 32, 1, 9, "constant", "sigmoid", "4")]


@pytest.mark.parametrize("case", MARCH, ids=lambda c: "x".join(map(str, c[:6])) + "_" + c[6] + "_segs" + (c[8] or "auto"))
def test_fp16_rowfold_marching_strips_match_the_tile_kernel_and_the_oracle(ctx, monkeypatch, case):
    """conv2d_rowmarch.hip: several iterations of the LDS row ring per block, several row segments per strip (forced and chosen), ragged last iteration and
    last strip, odd widths (rows that start on a 2-byte boundary take the 2-byte store path) -- against the quantised oracle and conv2d_rowfold.hip's tile
    kernel (SNNHIP_ROWFOLD=tile) on the same inputs."""
    N, H, W, IC, OC, k, pad_mode, act, segs = case
    x = _rand((N, H, W, IC), 301)
    w = _rand((OC, IC, k, k), 302, 1.0 / np.sqrt(IC * k * k))
    b = _rand((OC,), 303, 0.1)
    bn = _bn(OC, 304) if act else None
    pads = O.padding_offsets("same", k)
    if segs:
        monkeypatch.setenv("SNNHIP_ROWFOLD_SEGS", segs)
    y, desc = _conv16(ctx, x, w, b, 1, pads, pad_mode, act, bn)
    assert "conv2d_rowfold" in desc and "row-marching" in desc and (not segs or "segments=%s x" % segs in desc), desc
    want = O._h(O.conv2d(O._h(x), O._h(w), b, 1, pads, pad_mode, act, 0.0, bn))
    np.testing.assert_allclose(y, want, err_msg=desc, **TOLH)
    monkeypatch.setenv("SNNHIP_ROWFOLD", "tile")
    y2, desc2 = _conv16(ctx, x, w, b, 1, pads, pad_mode, act, bn)
    assert "conv2d_rowfold" in desc2 and "row-marching" not in desc2, desc2
    np.testing.assert_allclose(y, y2, err_msg=desc + " vs " + desc2, rtol=2e-3, atol=2e-3)


S2M = [(1, 40, 70, 32, 64, "constant", "relu", ""), (2, 33, 61, 64, 128, "reflect", "", "2"), (1, 21, 130, 32, 128, "replicate", "leakyRelu", ""),
       (1, 50, 37, 64, 256, "constant", "tanh", "3"), (2, 9, 9, 32, 64, "constant", "", ""), (1, 64, 64, 64, 128, "constant", "relu6", "")]


@pytest.mark.parametrize("case", S2M, ids=lambda c: "x".join(map(str, c[:5])) + "_" + c[5] + "_segs" + (c[7] or "auto"))
def test_fp16_stride2_marching_strips_match_the_oracle_and_the_tile_kernel(ctx, monkeypatch, case):
    """conv2d_s2march.hip (forced: the default takes it on large maps only): ragged strips and last iterations, several row segments, several
    output-channel blocks per strip (OC > 64 / 128), every padding mode, BN + activations -- against the quantised oracle and the 128-pixel kernel."""
    N, H, W, IC, OC, pad_mode, act, segs = case
    x = _rand((N, H, W, IC), 401)
    w = _rand((OC, IC, 3, 3), 402, 1.0 / np.sqrt(IC * 9))
    b = _rand((OC,), 403, 0.1)
    bn = _bn(OC, 404) if act else None
    pads = O.padding_offsets("same", 3)
    monkeypatch.setenv("SNNHIP_CONV", "s2march")
    if segs:
        monkeypatch.setenv("SNNHIP_S2MARCH_SEGS", segs)
    y, desc = _conv16(ctx, x, w, b, 2, pads, pad_mode, act, bn)
    assert "s=2" in desc and "row-marching" in desc and (not segs or "segments=%s x" % segs in desc), desc
    want = O._h(O.conv2d(O._h(x), O._h(w), b, 2, pads, pad_mode, act, 0.0, bn))
    assert y.shape == want.shape
    np.testing.assert_allclose(y, want, err_msg=desc, **TOLH)
    monkeypatch.setenv("SNNHIP_CONV", "mfma")
    y2, desc2 = _conv16(ctx, x, w, b, 2, pads, pad_mode, act, bn)
    assert "row-marching" not in desc2, desc2
    np.testing.assert_allclose(y, y2, err_msg=desc + " vs " + desc2, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("n,h,w,ic,oc,act", [(1, 40, 56, 32, 64, "relu"), (2, 37, 45, 64, 128, "relu"), (1, 25, 80, 32, 64, "leakyRelu")])
def test_fp16_stride2_marching_with_fused_pad_and_instancenorm(ctx, monkeypatch, n, h, w, ic, oc, act):
    """Graph rules D + I on the stride-2 strips: InstanceNorm -> reflect Pad -> Conv2D 3x3 stride 2 as the norm's statistics sweep + ONE convolution
    launch that resolves the pad in its row / column look-ups and normalises what it stages: bit-identical to the separate launches."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_CONV", "s2march")
    x = 1.5 * _rand((n, h, w, ic), 1) + 0.2
    wt, b = _rand((oc, ic, 3, 3), 2, 1.0 / np.sqrt(ic * 9)), _rand((oc,), 3, 0.5)
    beta, gamma = _rand((ic,), 4, 0.3), 1.0 + _rand((ic,), 5, 0.2)
    norm = snn.instancenorm_plan(ctx, n, h, w, ic, beta, gamma, act=act, leaky=0.1)
    pad = snn.pad_plan(ctx, n, h, w, ic, (1, 1, 1, 1), "reflect")
    conv = snn.conv2d_plan(ctx, n, h + 2, w + 2, wt, b, stride=2, pads=(0, 0, 0, 0), act="relu", dtype=snn.F16)
    assert "row-marching" in conv.describe(), conv.describe()
    plans = [norm, pad, conv]
    fused = snn.chain_plan(ctx, plans)
    d = fused.describe()
    assert fused.num_steps() == 1 and "in the staging" in d and "row-marching" in d and "+pad(reflect)" in d, d
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    y = fused(xt).numpy()
    t = xt
    for pl in plans:
        t = pl(t)
    np.testing.assert_array_equal(y, t.numpy())
    ref = O.pad(O._h(O.instancenorm(O._h(x), beta, gamma, act, 0.1)), (1, 1, 1, 1), "reflect")
    want = O._h(O.conv2d(ref, O._h(wt), b, 2, (0, 0, 0, 0), "constant", "relu", 0.0, None))
    np.testing.assert_allclose(y, want, err_msg=d, rtol=6e-3, atol=6e-3)


def test_fp16_rowfold_with_fused_reflect_pad(ctx):
    """Chain rule D on the row-fold kernel: Pad(reflect 4) -> Conv2D 9x9 "valid" 32 -> 3 (Candy's output layer) as one launch."""
    import shadernn_amd as snn

    N, H, W = 2, 30, 50
    x = _rand((N, H, W, 32), 211)
    w, b = _rand((3, 32, 9, 9), 212, 1.0 / np.sqrt(32 * 81)), _rand((3,), 213, 0.1)
    pad = snn.pad_plan(ctx, N, H, W, 32, (4, 4, 4, 4), "reflect")
    conv = snn.conv2d_plan(ctx, N, H + 8, W + 8, w, b, pads=(0, 0, 0, 0), dtype=snn.F16)
    assert "conv2d_rowfold" in conv.describe()
    chain = snn.chain_plan(ctx, [pad, conv])
    assert chain.num_steps() == 1 and "rowfold" in chain.describe() and "+pad(reflect)" in chain.describe(), chain.describe()
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    got = chain(xt).numpy()
    sep = conv(pad(xt)).numpy()
    want = O._h(O.conv2d(O.pad(O._h(x), (4, 4, 4, 4), "reflect"), O._h(w), b, 1, (0, 0, 0, 0), "constant", ""))
    assert got.shape == want.shape  # "valid" keeps the padded extent (Q20)
    np.testing.assert_allclose(got, want, **TOLH)
    np.testing.assert_array_equal(got, sep)


def test_fp16_stride2_marching_block_statistics_feed_the_instancenorm_behind_it(ctx, monkeypatch):
    """Rule F on conv2d_s2march.hip (32 input channels): Conv2D 3x3 stride 2 -> InstanceNorm as the convolution (block records + in-kernel fold) + ONE
    normalise sweep; against the separate launches and the oracle, two different inputs through the same plan."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_CONV", "s2march")
    monkeypatch.setenv("SNNHIP_NORM_FUSION_MIN_MB", "0")
    monkeypatch.setenv("SNNHIP_S2MARCH_SEGS", "2")
    n, h, w, ic, oc = 2, 61, 90, 32, 64
    wt, b = _rand((oc, ic, 3, 3), 2, 1.0 / np.sqrt(ic * 9)), _rand((oc,), 3, 0.5) + 2.0
    beta, gamma = _rand((oc,), 4, 0.3), 1.0 + _rand((oc,), 5, 0.2)
    conv = snn.conv2d_plan(ctx, n, h, w, wt, b, stride=2, pads=O.padding_offsets("same", 3), act="", dtype=snn.F16)
    oh, ow = conv.out_shape()[1:3]
    norm = snn.instancenorm_plan(ctx, n, oh, ow, oc, beta, gamma, act="relu")
    fused = snn.chain_plan(ctx, [conv, norm])
    d = fused.describe()
    assert fused.num_steps() == 1 and "row-marching" in d and "+tile-stats+fold" in d and "instancenorm(1 sweep)" in d, d
    for seed in (1, 11):
        x = _rand((n, h, w, ic), seed) if seed == 1 else (0.5 * _rand((n, h, w, ic), seed) + 0.75).astype(np.float32)
        xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
        y = fused(xt).numpy()
        np.testing.assert_array_equal(fused(xt).numpy(), y)
        np.testing.assert_allclose(y, norm(conv(xt)).numpy(), rtol=2e-3, atol=2e-3, err_msg="seed %d: %s" % (seed, d))


@pytest.mark.parametrize("shape", [(2, 50, 70), (3, 33, 100), (1, 16, 32)], ids=["2x50x70", "3x33x100_many_blocks_per_image", "one_tile"])
def test_fp16_rgb_stem_block_statistics_feed_the_instancenorm_behind_it(ctx, monkeypatch, shape):
    """Rule F on conv2d_stem_f16.hip: [Pad ->] Conv2D 9x9 (3 -> 32) -> InstanceNorm as the convolution (one record per persistent block and image, folded by
    the block that counts in last) + ONE normalise sweep; against the separate launches, two different inputs through the same plan (the counters re-arm)."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_NORM_FUSION_MIN_MB", "0")
    n, h, w = shape
    wt, b = _rand((32, 3, 9, 9), 2, 1.0 / np.sqrt(3 * 81)), _rand((32,), 3, 0.3) + 0.5
    beta, gamma = _rand((32,), 4, 0.3), 1.0 + _rand((32,), 5, 0.2)
    pad = snn.pad_plan(ctx, n, h, w, 3, (4, 4, 4, 4), "reflect")
    conv = snn.conv2d_plan(ctx, n, h + 8, w + 8, wt, b, pads=(0, 0, 0, 0), act="relu", dtype=snn.F16)
    assert "conv2d_mfma_stem_f16" in conv.describe(), conv.describe()
    oh, ow = conv.out_shape()[1:3]
    norm = snn.instancenorm_plan(ctx, n, oh, ow, 32, beta, gamma, act="relu")
    fused = snn.chain_plan(ctx, [pad, conv, norm])
    d = fused.describe()
    assert fused.num_steps() == 1 and "stem_f16" in d and "+tile-stats+fold" in d and "instancenorm(1 sweep)" in d and "+pad(reflect)" in d, d
    for seed in (1, 11):
        x = _rand((n, h, w, 3), seed) if seed == 1 else (0.5 * _rand((n, h, w, 3), seed) + 0.75).astype(np.float32)
        xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
        y = fused(xt).numpy()
        np.testing.assert_array_equal(fused(xt).numpy(), y)
        np.testing.assert_allclose(y, norm(conv(pad(xt))).numpy(), rtol=2e-3, atol=2e-3, err_msg="seed %d: %s" % (seed, d))
    monkeypatch.setenv("SNNHIP_STEM_NO_STATS", "1")
    d2 = snn.chain_plan(ctx, [pad, conv, norm]).describe()
    assert "tile-stats" not in d2, d2
