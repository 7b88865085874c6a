"""GPU parity tests of the fp32-MFMA implicit-GEMM convolution (conv2d_mfma.hip) vs the CPU oracle, through the C-ABI.
SNNHIP_CONV=mfma|generic pins the routing so both kernels are checked on the same inputs (and against each other)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from test_ops_gpu import TOL, _bn, _rand, run_conv

pytestmark = pytest.mark.gpu


@pytest.fixture
def force():
    old = os.environ.get("SNNHIP_CONV")

    def _set(v):
        if v is None:
            os.environ.pop("SNNHIP_CONV", None)
        else:
            os.environ["SNNHIP_CONV"] = v

    yield _set
    _set(old)


MFMA_CASES = [
    # (N, H, W, IC, OC, k, stride)            shape class
    (2, 56, 56, 64, 64, 3, 1),    # ResNet-18 layer1
    (2, 28, 28, 128, 128, 3, 1),  # ResNet-18 layer2
    (2, 56, 56, 64, 128, 3, 2),   # ResNet-18 layer2 first conv (stride 2, 128-wide N block)
    (2, 56, 56, 64, 128, 1, 2),   # ResNet-18 downsample 1x1 s2
    (3, 14, 14, 256, 256, 3, 1),  # layer3: tile spans 14x14
    (5, 7, 7, 512, 512, 3, 1),    # layer4: several images per pixel tile
    (1, 30, 30, 3, 64, 7, 2),     # stem (IC=3, forced: one 8-channel chunk mostly padding)
    (2, 28, 28, 32, 192, 1, 1),   # MobileNetV2 expand
    (2, 28, 28, 192, 32, 1, 1),   # MobileNetV2 project (N block 32)
    (1, 14, 14, 96, 576, 1, 1), (1, 7, 7, 320, 1280, 1, 1),
    (1, 17, 23, 24, 144, 1, 1),   # ragged, OC not a multiple of 32, IC not a multiple of 16
    (2, 9, 11, 20, 33, 3, 1),     # IC%16 != 0, OC%32 != 0
    (1, 12, 12, 10, 18, 3, 2),    # IC not a multiple of 4: scalar staging path
    (1, 20, 24, 32, 32, 5, 1), (1, 16, 20, 16, 16, 9, 1),  # 5x5 and 9x9 (Candy)
    (1, 9, 9, 8, 16, 4, 1),       # even kernel, asymmetric padding; IC == 8 -> one 8-channel chunk
    (1, 40, 130, 16, 16, 3, 1),   # wide image: 1x1x128 / 1x2x64 pixel tiles
]


@pytest.mark.parametrize("case", MFMA_CASES, ids=lambda c: "x".join(map(str, c)))
def test_mfma_conv_matches_oracle(ctx, force, case):
    N, H, W, IC, OC, k, s = case
    x = _rand((N, H, W, IC), 41)
    w = _rand((OC, IC, k, k), 42, 1.0 / np.sqrt(IC * k * k))
    b = _rand((OC,), 43, 0.1)
    bn = _bn(OC, 44)
    pads = O.padding_offsets("same", k)
    force("mfma")
    y, desc = run_conv(ctx, x, w, b, s, pads, "constant", "relu", 0.0, bn)
    assert "mfma" in desc, desc
    want = O.conv2d(x, w, b, s, pads, "constant", "relu", 0.0, bn)
    assert y.shape == want.shape, desc
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)
    force("generic")
    yg, descg = run_conv(ctx, x, w, b, s, pads, "constant", "relu", 0.0, bn)
    assert "generic" in descg, descg
    np.testing.assert_allclose(y, yg, err_msg=desc + " vs " + descg, **TOL)


@pytest.mark.parametrize("pad_mode", ["constant", "replicate", "reflect", "none"])
@pytest.mark.parametrize("act", ["", "relu", "relu6", "tanh", "sigmoid", "leakyRelu", "SiLU", "SiLU_quirk"])
def test_mfma_conv_padding_modes_activations(ctx, force, pad_mode, act):
    x = _rand((2, 13, 18, 24), 45)
    w = _rand((40, 24, 3, 3), 46, 0.1)
    b = _rand((40,), 47, 0.1)
    bn = _bn(40, 48)
    force("mfma")
    y, desc = run_conv(ctx, x, w, b, 1, (1, 1, 1, 1), pad_mode, act, 0.1, bn)
    assert "mfma" in desc, desc
    want = O.conv2d(x, w, b, 1, (1, 1, 1, 1), pad_mode, act, 0.1, bn)
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)


def test_default_routing(ctx, force):
    """Layers with >= 32 output channels (or >= 16 with >= 32 input channels) go to the MFMA kernel, even IC = 3 (measured faster);
    channel-thin ones to the VALU kernel."""
    force(None)
    for (ic, oc, want) in [(64, 64, "mfma"), (3, 64, "mfma"), (16, 4, "thin"), (128, 1, "thin"), (3, 3, "generic"), (16, 16, "generic"), (64, 16, "mfma"), (8, 32, "mfma")]:
        x = _rand((1, 8, 8, ic), 1)
        w = _rand((oc, ic, 3, 3), 2, 0.1)
        _, desc = run_conv(ctx, x, w, None, 1, (1, 1, 1, 1), "constant", "relu", 0.0, None)
        assert want in desc, (ic, oc, desc)


def test_mfma_conv_linearity_at_scale(ctx, force):
    """Size-independent property at a ResNet-18 batch-32 layer size (too big for the oracle in seconds):
    conv(a*x1 + x2) == a*conv(x1) + conv(x2) with no bias/activation, and a sampled window equals the oracle."""
    import shadernn_amd as snn

    force("mfma")
    N, H, W, IC, OC = 32, 56, 56, 64, 64
    x1 = _rand((N, H, W, IC), 51)
    x2 = _rand((N, H, W, IC), 52)
    w = _rand((OC, IC, 3, 3), 53, 1.0 / 24.0)
    plan = snn.conv2d_plan(ctx, N, H, W, w, None, stride=1, pads=(1, 1, 1, 1), pad_mode="constant", act="", leaky=0.0, bn=None)
    assert "mfma" in plan.describe()

    def run(x):
        xt = snn.Tensor.from_numpy(ctx, x)
        yt = plan(xt)
        y = yt.numpy()
        xt.free()
        yt.free()
        return y

    y1, y2, y3 = run(x1), run(x2), run(2.5 * x1 + x2)
    np.testing.assert_allclose(y3, 2.5 * y1 + y2, rtol=1e-4, atol=2e-4)
    # window: image 17, rows 20..27 == interior rows 2..9 of the oracle run on input rows 18..29
    want = O.conv2d(x1[17:18, 18:30], w, None, 1, (1, 1, 1, 1), "constant", "")
    np.testing.assert_allclose(y1[17:18, 20:28], want[:, 2:10], **TOL)
    plan.destroy()


@pytest.mark.parametrize("case,split", [((5, 7, 7, 512, 512, 3, 1), None), ((2, 14, 14, 256, 256, 3, 1), None), ((2, 9, 11, 80, 33, 3, 1), "3"),
                                        ((1, 7, 7, 320, 1280, 1, 1), "4"), ((2, 28, 28, 64, 128, 3, 2), "2")])
def test_mfma_conv_split_k(ctx, force, monkeypatch, case, split):
    """Split-K (deep K, few output tiles): partial sums + deterministic reduce pass give the same result as the oracle; the default
    heuristic turns it on for the ResNet 7x7 / 14x14 stages."""
    N, H, W, IC, OC, k, s = case
    if split is not None:
        monkeypatch.setenv("SNNHIP_CONV_SPLITK", split)
    x = _rand((N, H, W, IC), 61)
    w = _rand((OC, IC, k, k), 62, 1.0 / np.sqrt(IC * k * k))
    b = _rand((OC,), 63, 0.1)
    bn = _bn(OC, 64)
    pads = O.padding_offsets("same", k)
    force("mfma")
    for act in ("relu", "tanh"):
        y, desc = run_conv(ctx, x, w, b, s, pads, "constant", act, 0.0, bn)
        assert "splitK=1" not in desc, desc
        want = O.conv2d(x, w, b, s, pads, "constant", act, 0.0, bn)
        np.testing.assert_allclose(y, want, err_msg=desc, **TOL)
    y2, _ = run_conv(ctx, x, w, b, s, pads, "constant", "tanh", 0.0, bn)
    np.testing.assert_array_equal(y, y2)  # deterministic


THIN_CASES = [(1, 40, 70, 32, 3, 9, 1), (2, 20, 40, 16, 4, 3, 1), (1, 17, 33, 24, 1, 5, 1), (1, 9, 9, 10, 2, 3, 1), (3, 5, 7, 64, 4, 1, 1), (1, 33, 65, 8, 3, 7, 1)]


@pytest.mark.parametrize("case", THIN_CASES, ids=lambda c: "x".join(map(str, c)))
def test_thin_conv_matches_oracle(ctx, force, case):
    """OC <= 4 layers on v_mfma_f32_4x4x1 (conv2d_thin.hip): Candy's 9x9 32->3 output conv, ESPCN's unfused 16->4, ..."""
    N, H, W, IC, OC, k, s = case
    x = _rand((N, H, W, IC), 71)
    w = _rand((OC, IC, k, k), 72, 1.0 / np.sqrt(IC * k * k))
    b = _rand((OC,), 73, 0.1)
    bn = _bn(OC, 74)
    pads = O.padding_offsets("same", k)
    force("thin")
    for pad_mode, act in (("constant", "tanh"), ("reflect", "relu"), ("replicate", "SiLU_quirk"), ("none", "")):
        y, desc = run_conv(ctx, x, w, b, s, pads, pad_mode, act, 0.0, bn)
        assert "thin" in desc, desc
        want = O.conv2d(x, w, b, s, pads, pad_mode, act, 0.0, bn)
        np.testing.assert_allclose(y, want, err_msg=desc + " " + pad_mode + " " + act, **TOL)
    force(None)
    _, desc = run_conv(ctx, x, w, b, s, pads, "constant", "relu", 0.0, None)
    assert ("thin" in desc) == (IC >= 8), desc


PAIR_CASES = [
    # (N, H, W, IC, OC, k, stride): channel-thin inputs -> the tap-pair K loop (two taps per MFMA K step)
    (2, 37, 41, 3, 64, 7, 2),   # ResNet stem: 49 taps (odd: the last h=1 half is zero weights)
    (1, 40, 56, 3, 32, 3, 2),   # MobileNetV2 stem
    (1, 33, 47, 3, 16, 3, 1),   # YOLOv3-tiny first conv (OC 16: half of the 32-wide block is padding)
    (1, 24, 40, 3, 32, 9, 1),   # Candy conv1 (81 taps)
    (1, 21, 19, 4, 48, 4, 1),   # even kernel, IC == 4: vector staging
    (1, 18, 22, 1, 32, 5, 1),   # single channel, 5x5
    (2, 16, 16, 2, 40, 2, 2),   # 2x2 stride 2: two steps
    (1, 15, 31, 3, 64, 1, 1),   # 1x1 stays on the channel-chunk path (one tap)
]


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
@pytest.mark.parametrize("case", PAIR_CASES, ids=lambda c: "x".join(map(str, c)))
def test_tap_pair_mode_matches_oracle(ctx, force, monkeypatch, case, pad_mode, dtype):
    import shadernn_amd as snn

    N, H, W, IC, OC, k, s = case
    x = _rand((N, H, W, IC), 51)
    w = _rand((OC, IC, k, k), 52, 1.0 / np.sqrt(IC * k * k))
    b = _rand((OC,), 53, 0.1)
    pads = O.padding_offsets("same", k)
    force("mfma")
    dt = snn.F16 if dtype == "f16" else snn.F32

    def run():
        plan = snn.conv2d_plan(ctx, N, H, W, w, b, stride=s, pads=pads, pad_mode=pad_mode, act="relu", dtype=dt)
        xt = snn.Tensor.from_numpy(ctx, x, dtype=dt)
        yt = plan(xt)
        out, desc = yt.numpy(), plan.describe()
        xt.free(), yt.free(), plan.destroy()
        return out, desc

    y, desc = run()
    assert ("tap-pairs" in desc) == (k > 1), desc
    if dtype == "f16":
        want = O._h(O.conv2d(O._h(x), O._h(w), b, s, pads, pad_mode, "relu", 0.0, None))
        np.testing.assert_allclose(y, want, err_msg=desc, rtol=2e-3, atol=2e-3)
    else:
        np.testing.assert_allclose(y, O.conv2d(x, w, b, s, pads, pad_mode, "relu", 0.0, None), err_msg=desc, **TOL)
    monkeypatch.setenv("SNNHIP_CONV_PAIR", "0")  # the channel-chunk path on the same inputs
    y2, desc2 = run()
    assert "tap-pairs" not in desc2
    np.testing.assert_allclose(y, y2, err_msg=desc + " vs " + desc2, rtol=2e-3 if dtype == "f16" else 1e-4, atol=2e-3 if dtype == "f16" else 1e-4)


STREAM_CASES = [
    # (N, H, W, IC, OC, act, bn)   pointwise stride-1 layers through conv1x1_stream.hip (SNNHIP_CONV_1X1=2 lifts its size threshold)
    (2, 28, 28, 16, 96, "relu6", True),     # MobileNetV2 expand: 96-wide blocks
    (2, 28, 28, 24, 144, "relu6", True),    # IC not a multiple of 16, OC padded 144 -> 192
    (1, 33, 47, 32, 192, "relu", False),    # ragged pixel count (1551 = 48 tiles + 15 rows)
    (2, 14, 14, 96, 24, "", True),          # project: one 32-wide block, 8 of its columns padding
    (1, 19, 21, 144, 32, "", True), (1, 12, 12, 192, 64, "tanh", False),
    (1, 9, 9, 8, 16, "leakyRelu", True),    # smallest supported: one chunk, 16 channels
    (1, 10, 10, 40, 576, "SiLU", True),     # 6 channel blocks, non-simple activation
    (3, 5, 7, 64, 100, "sigmoid", False),   # OC % 32 != 0 with 64-wide... (100 -> 128: falls to the 32-wide rule)
]


@pytest.mark.parametrize("case", STREAM_CASES, ids=lambda c: "x".join(map(str, c)))
def test_pointwise_stream_matches_oracle(ctx, monkeypatch, case):
    N, H, W, IC, OC, act, use_bn = case
    x = _rand((N, H, W, IC), 51)
    w = _rand((OC, IC, 1, 1), 52, 1.0 / np.sqrt(IC))
    b = _rand((OC,), 53, 0.1)
    bn = _bn(OC, 54) if use_bn else None
    monkeypatch.setenv("SNNHIP_CONV_1X1", "2")
    y, desc = run_conv(ctx, x, w, b, 1, (0, 0, 0, 0), "constant", act, 0.1, bn)
    assert "stream" in desc, desc
    want = O.conv2d(x, w, b, 1, (0, 0, 0, 0), "constant", act, 0.1, bn)
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)
    monkeypatch.setenv("SNNHIP_CONV_1X1", "0")  # the general kernel on the same layer
    y2, desc2 = run_conv(ctx, x, w, b, 1, (0, 0, 0, 0), "constant", act, 0.1, bn)
    assert "stream" not in desc2, desc2
    np.testing.assert_allclose(y, y2, err_msg=desc + " vs " + desc2, **TOL)


@pytest.mark.parametrize("case", [(2, 56, 56, 64, 128), (1, 27, 31, 32, 64), (3, 14, 14, 256, 512)], ids=lambda c: "x".join(map(str, c)))
def test_pointwise_stream_stride2(ctx, monkeypatch, case):
    """ResNet's 1x1 stride-2 downsample convolutions: the stream kernel decodes (n, 2*oy, 2*ox) per lane; odd sizes round the output up."""
    N, H, W, IC, OC = case
    x = _rand((N, H, W, IC), 71)
    w = _rand((OC, IC, 1, 1), 72, 1.0 / np.sqrt(IC))
    b = _rand((OC,), 73, 0.1)
    bn = _bn(OC, 74)
    monkeypatch.setenv("SNNHIP_CONV_1X1", "2")
    y, desc = run_conv(ctx, x, w, b, 2, (0, 0, 0, 0), "constant", "", 0.0, bn)
    assert "stream" in desc and "s=2" in desc, desc
    want = O.conv2d(x, w, b, 2, (0, 0, 0, 0), "constant", "", 0.0, bn)
    assert y.shape == want.shape, (y.shape, want.shape, desc)
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)


def test_pointwise_stream_full_size_and_fused_add(ctx):
    """At MobileNetV2's own size the kernel is the default route; with an Add behind it (chain rule E) the residual goes through its epilogue."""
    import shadernn_amd as snn

    n, h, w, ic, oc = 16, 56, 56, 144, 24
    x, wt, b = _rand((n, h, w, ic), 61), _rand((oc, ic, 1, 1), 62, 1.0 / np.sqrt(ic)), _rand((oc,), 63, 0.1)
    skip, bn = _rand((n, h, w, oc), 64), _bn(oc, 65)
    conv = snn.conv2d_plan(ctx, n, h, w, wt, b, stride=1, pads=(0, 0, 0, 0), act="", bn=bn)
    assert "stream" in conv.describe(), conv.describe()
    add = snn.add_plan(ctx, n, h, w, oc, act="relu")
    fused = snn.chain_plan(ctx, [conv, add])
    assert fused.num_steps() == 1 and "stream" in fused.describe() and "+add" in fused.describe(), fused.describe()
    xt, st = snn.Tensor.from_numpy(ctx, x), snn.Tensor.from_numpy(ctx, skip)
    y = fused([xt, st]).numpy()
    want = O.add_act(O.conv2d(x, wt, b, 1, (0, 0, 0, 0), "constant", "", 0.0, bn), skip, "relu", 0.0)
    np.testing.assert_allclose(y, want, err_msg=fused.describe(), **TOL)
    np.testing.assert_allclose(y, add([conv(xt), st]).numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("oc,with_add", [(160, True), (320, False), (36, False)])
def test_pointwise_stream_weight_slice_in_k_phases(ctx, monkeypatch, oc, with_add):
    """MobileNetV2's 960-channel pointwise layers of the 7x7 stage at batch 256 (12 544 pixel rows): the [960][32] weight slice does not fit the LDS
    budget and is staged in two K phases (barriers between them, an idle wave in the last block, the activation prefetch running across the phase
    boundary); ragged last pixel tile, ragged last column block, fused residual Add.  Against the oracle and the general kernel."""
    import shadernn_amd as snn

    n, h, w, ic = 8, 32, 32 + 1, 960  # 8448 pixel rows (>= 8192), not a multiple of 128
    x, wt, b = _rand((n, h, w, ic), 71), _rand((oc, ic, 1, 1), 72, 1.0 / np.sqrt(ic)), _rand((oc,), 73, 0.1)
    bn = _bn(oc, 74)
    conv = snn.conv2d_plan(ctx, n, h, w, wt, b, stride=1, pads=(0, 0, 0, 0), act="relu6" if not with_add else "", bn=bn)
    assert "stream" in conv.describe() and "K phases" in conv.describe() and "in 1 K phases" not in conv.describe(), conv.describe()
    xt = snn.Tensor.from_numpy(ctx, x)
    want = O.conv2d(x, wt, b, 1, (0, 0, 0, 0), "constant", "relu6" if not with_add else "", 0.0, bn, threads=8)
    if with_add:
        skip = _rand((n, h, w, oc), 75)
        add = snn.add_plan(ctx, n, h, w, oc, act="")
        fused = snn.chain_plan(ctx, [conv, add])
        assert fused.num_steps() == 1 and "K phases" in fused.describe() and "+add" in fused.describe(), fused.describe()
        y = fused([xt, snn.Tensor.from_numpy(ctx, skip)]).numpy()
        want = O.add_act(want, skip, "", 0.0)
    else:
        y = conv(xt).numpy()
    np.testing.assert_allclose(y, want, err_msg=conv.describe(), **TOL)
    monkeypatch.setenv("SNNHIP_CONV_1X1_PHASES", "0")
    gen = snn.conv2d_plan(ctx, n, h, w, wt, b, stride=1, pads=(0, 0, 0, 0), act="relu6" if not with_add else "", bn=bn)
    assert "K phases" not in gen.describe(), gen.describe()
    if not with_add:
        np.testing.assert_allclose(y, gen(xt).numpy(), rtol=2e-5, atol=2e-5)


MARCH_CASES = [
    # (N, H, W, IC, OC, act, bn)   widening pointwise layers through conv1x1_march_kernel (SNNHIP_CONV_1X1_MARCH=1 forces it on layers this small)
    (4, 14, 14, 64, 384, "relu6", True),     # MobileNetV2 64 -> 384: 9 DMA pieces per tile, four tiles per step, four output blocks
    (2, 14, 14, 96, 576, "relu6", True),     # 96 -> 576: 13 pieces, six output blocks, 13 tiles = ragged steps (392 px: the last tile 8 rows)
    (8, 7, 7, 160, 960, "relu6", True),      # 160 -> 960: 21 pieces, two tiles per step, ten output blocks
    (1, 33, 47, 32, 96, "relu", False),      # 5 pieces; 1551 pixels = 48 tiles + 15 rows
    (2, 9, 11, 128, 384, "SiLU", True),      # 17 pieces, non-simple activation
    (1, 5, 5, 16, 96, "", False),            # one tile, one step
    (3, 20, 20, 24, 96, "leakyRelu", True),  # IC not a multiple of 16: 7 slots per row
]


@pytest.mark.parametrize("case", MARCH_CASES, ids=lambda c: "x".join(map(str, c)))
def test_pointwise_march_matches_oracle_and_the_stream_kernel(ctx, monkeypatch, case):
    """conv1x1_march_kernel: persistent blocks (weight slice resident in LDS), one loader wave filling an LDS ring of activation tiles by LDS-DMA, 3 x TS compute
    waves (tile x 32 channels each).  Whole output against the oracle and, bit for bit, against conv1x1_stream_kernel on the same layer (same MFMA order)."""
    N, H, W, IC, OC, act, use_bn = case
    x = _rand((N, H, W, IC), 61)
    w = _rand((OC, IC, 1, 1), 62, 1.0 / np.sqrt(IC))
    b = _rand((OC,), 63, 0.1)
    bn = _bn(OC, 64) if use_bn else None
    monkeypatch.setenv("SNNHIP_CONV_1X1", "2")
    monkeypatch.setenv("SNNHIP_CONV_1X1_MARCH", "1")
    y, desc = run_conv(ctx, x, w, b, 1, (0, 0, 0, 0), "constant", act, 0.1, bn)
    assert "march: persistent blocks" in desc and "conv1x1_march_kernel" in desc, desc
    want = O.conv2d(x, w, b, 1, (0, 0, 0, 0), "constant", act, 0.1, bn)
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)
    y_again, _ = run_conv(ctx, x, w, b, 1, (0, 0, 0, 0), "constant", act, 0.1, bn)
    np.testing.assert_array_equal(y, y_again)
    monkeypatch.setenv("SNNHIP_CONV_1X1_MARCH", "0")
    y2, desc2 = run_conv(ctx, x, w, b, 1, (0, 0, 0, 0), "constant", act, 0.1, bn)
    assert "stream: wave" in desc2 and "march" not in desc2, desc2
    np.testing.assert_array_equal(y, y2, err_msg=desc + " vs " + desc2)


def test_pointwise_march_is_the_default_on_large_widening_layers_only(ctx, monkeypatch):
    import shadernn_amd as snn

    monkeypatch.delenv("SNNHIP_CONV_1X1_MARCH", raising=False)
    monkeypatch.delenv("SNNHIP_CONV_1X1", raising=False)

    def kind(N, H, W, IC, OC, stride=1):
        wgt = _rand((OC, IC, 1, 1), 1)
        return snn.conv2d_plan(ctx, N, H, W, wgt, None, stride=stride, pads=(0, 0, 0, 0), act="relu6").describe()

    assert "march" in kind(256, 14, 14, 96, 576)         # MobileNetV2 b11 at batch 256
    assert "march" in kind(256, 14, 14, 64, 384)
    assert "march" not in kind(4, 14, 14, 96, 576)       # a few tiles per CU: the one-tile-per-wave kernel
    assert "march" not in kind(256, 14, 14, 576, 96)     # a project layer (narrowing)
    assert "march" not in kind(256, 14, 14, 96, 160)     # not whole 96-channel blocks
