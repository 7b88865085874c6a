"""GPU parity tests: every operator through the C-ABI vs the CPU oracle on identical seeded inputs.
Tolerance = the north-star's 1e-4 (fp32), tighter than the reference's own 1e-2 (convolutionTest.cpp:158)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-4, atol=1e-4)


def _rand(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def _bn(c, seed):
    r = np.random.default_rng(seed)
    return {"beta": r.uniform(-0.1, 0.1, c).astype(np.float32), "gamma": r.uniform(0.5, 1.5, c).astype(np.float32),
            "mean": r.uniform(-0.1, 0.1, c).astype(np.float32), "var": r.uniform(0.5, 1.5, c).astype(np.float32)}


def run_conv(ctx, x, w, b, stride, pads, pad_mode, act, leaky, bn, depthwise=False):
    import shadernn_amd as snn

    n, h, ww, _ = x.shape
    plan = snn.conv2d_plan(ctx, n, h, ww, w, b, stride=stride, pads=pads, pad_mode=pad_mode, act=act, leaky=leaky, bn=bn, depthwise=depthwise)
    xt = snn.Tensor.from_numpy(ctx, x)
    yt = plan(xt)
    y = yt.numpy()
    desc = plan.describe()
    for t in (xt, yt):
        t.free()
    plan.destroy()
    return y, desc


CONV_CASES = [
    # (N, H, W, IC, OC, k, stride)
    (1, 8, 8, 128, 1, 1, 1),     # convolutionTest.cpp defaults (:419-451)
    (1, 8, 8, 3, 1, 3, 1), (1, 8, 8, 4, 4, 3, 1), (1, 8, 8, 5, 5, 3, 2), (1, 8, 8, 16, 64, 3, 1),
    (2, 17, 23, 3, 7, 3, 1),     # ragged sizes, batch
    (1, 33, 70, 1, 16, 5, 1),    # ESPCN conv1 shape class
    (1, 20, 40, 16, 16, 3, 1),   # ESPCN conv2
    (1, 20, 40, 16, 4, 3, 1),    # ESPCN conv3
    (1, 30, 30, 3, 8, 7, 2),     # ResNet stem class
    (1, 14, 14, 32, 48, 1, 2),   # 1x1 stride 2 (ResNet downsample), masks instead of padding
    (1, 9, 9, 6, 10, 4, 1),      # even kernel: asymmetric padding (Q1)
    (3, 5, 5, 20, 33, 3, 1),     # tiles larger than the image
    (1, 12, 12, 3, 6, 9, 1),     # 9x9 (Candy)
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv2d_matches_oracle(ctx, case):
    N, H, W, IC, OC, k, s = case
    x = _rand((N, H, W, IC), 1)
    w = _rand((OC, IC, k, k), 2, 1.0 / np.sqrt(IC * k * k))
    b = _rand((OC,), 3, 0.1)
    pads = O.padding_offsets("same", k)
    y, desc = run_conv(ctx, x, w, b, s, pads, "constant", "relu", 0.0, None)
    want = O.conv2d(x, w, b, s, pads, "constant", "relu")
    assert y.shape == want.shape, desc
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)


@pytest.mark.parametrize("pad_mode", ["constant", "replicate", "reflect", "none"])
@pytest.mark.parametrize("act", ["", "relu", "relu6", "tanh", "sigmoid", "leakyRelu", "SiLU", "SiLU_quirk"])
def test_conv2d_padding_modes_activations_bn(ctx, pad_mode, act):
    x = _rand((1, 11, 13, 5), 4)
    w = _rand((9, 5, 3, 3), 5, 0.2)
    b = _rand((9,), 6, 0.1)
    bn = _bn(9, 7)
    y, desc = run_conv(ctx, x, w, b, 1, (1, 1, 1, 1), pad_mode, act, 0.1, bn)
    want = O.conv2d(x, w, b, 1, (1, 1, 1, 1), pad_mode, act, 0.1, bn)
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)


def test_conv2d_reference_unit_test_G1(ctx):
    """convolutionTest.cpp main(): 8x8x128 -> 1, k=1, input all ones, weights RandomMat after SRAND(7767517), bias 0,
    identity BN (still scales by 1/sqrt(1.001), quirk Q5)."""
    wts = O.reference_rand(7767517, 128).reshape(1, 128, 1, 1)
    x = np.ones((1, 8, 8, 128), np.float32)
    bn = {"beta": np.zeros(1, np.float32), "gamma": np.ones(1, np.float32), "mean": np.zeros(1, np.float32), "var": np.ones(1, np.float32)}
    y, desc = run_conv(ctx, x, wts, np.zeros(1, np.float32), 1, (0, 0, 0, 0), "constant", "", 0.0, bn)
    want = O.conv2d(x, wts, np.zeros(1, np.float32), 1, (0, 0, 0, 0), "constant", "", 0.0, bn)
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)
    np.testing.assert_allclose(y, np.full_like(y, wts.sum() / np.sqrt(np.float32(1.001))), rtol=1e-4)


@pytest.mark.parametrize("c,k,stride,hw", [(8, 1, 2, 9), (8, 3, 1, 9), (32, 3, 2, 15), (96, 3, 1, 12), (5, 3, 1, 7), (7, 5, 2, 11), (144, 3, 2, 14)])
def test_depthwise_matches_oracle(ctx, c, k, stride, hw):
    x = _rand((2, hw, hw + 3, c), 11)
    w = _rand((c, k, k), 12, 0.3)
    b = _rand((c,), 13, 0.1)
    bn = _bn(c, 14)
    pads = O.padding_offsets("same", k)
    y, desc = run_conv(ctx, x, w, b, stride, pads, "constant", "relu6", 0.0, bn, depthwise=True)
    want = O.depthwise(x, w, b, stride, pads, "relu6", 0.0, bn)
    assert y.shape == want.shape
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)


@pytest.mark.parametrize("act", ["relu", "", "sigmoid", "tanh", "softmax", "leakyRelu", "SiLU"])
@pytest.mark.parametrize("batch,inu,outu", [(1, 11, 5), (4, 512, 10), (2, 1280, 1000), (3, 37, 3)])
def test_dense_matches_oracle(ctx, act, batch, inu, outu):
    import shadernn_amd as snn

    x = _rand((batch, 1, 1, inu), 21)
    w = _rand((outu * inu,), 22, 1.0 / np.sqrt(inu))
    b = _rand((outu,), 23, 0.1)
    plan = snn.dense_plan(ctx, batch, w, outu, b, act=act, leaky=0.3)
    xt = snn.Tensor.from_numpy(ctx, x)
    yt = plan(xt)
    y = yt.numpy().reshape(batch, outu)
    want = O.dense(x.reshape(batch, inu), w, outu, b, act, 0.3)
    np.testing.assert_allclose(y, want, **TOL)


@pytest.mark.parametrize("act", ["relu", "", "softmax", "tanh"])
@pytest.mark.parametrize("batch,inu,outu", [(256, 1280, 1000), (32, 512, 1000), (33, 72, 37), (100, 40, 5), (32, 8, 32), (64, 32, 33), (32, 56, 7)])
def test_dense_batched_mfma_gemm_matches_oracle_and_the_wave_per_row_kernel(ctx, monkeypatch, act, batch, inu, outu):
    """dense_mfma_kernel (batch >= 32, fp32, In % 8 == 0): the classifier heads of BASELINE configs[2] / [3] at their batch sizes, ragged tiles in both
    directions, a K range that does not split evenly over the four waves, rows shorter than one 64-column staging chunk (in_units 8 / 32 / 56: the
    staging threads past the row must fall back to an address inside it); against the oracle and the wave-per-row kernel (SNNHIP_DENSE_MFMA=0)."""
    import shadernn_amd as snn

    x = _rand((batch, 1, 1, inu), 31)
    w = _rand((outu * inu,), 32, 1.0 / np.sqrt(inu))
    b = _rand((outu,), 33, 0.1)
    plan = snn.dense_plan(ctx, batch, w, outu, b, act=act, leaky=0.3)
    assert "GEMM" in plan.describe(), plan.describe()
    xt = snn.Tensor.from_numpy(ctx, x)
    y = plan(xt).numpy().reshape(batch, outu)
    want = O.dense(x.reshape(batch, inu), w, outu, b, act, 0.3)
    np.testing.assert_allclose(y, want, err_msg=plan.describe(), **TOL)
    monkeypatch.setenv("SNNHIP_DENSE_MFMA", "0")
    old = snn.dense_plan(ctx, batch, w, outu, b, act=act, leaky=0.3)
    assert "GEMM" not in old.describe()
    np.testing.assert_allclose(y, old(xt).numpy().reshape(batch, outu), rtol=2e-5, atol=2e-6)


def test_dense_flattens_hwc(ctx):
    """Dense consumes a [N,H,W,C] activation in HWC order (reference CPU flatten, cpulayer.h:94-113)."""
    import shadernn_amd as snn

    x = _rand((2, 3, 4, 6), 24)
    w = _rand((5 * 72,), 25, 0.1)
    plan = snn.dense_plan(ctx, 2, w, 5, None, act="")
    y = plan(snn.Tensor.from_numpy(ctx, x)).numpy().reshape(2, 5)
    np.testing.assert_allclose(y, O.dense(x.reshape(2, -1), w, 5, None, ""), **TOL)


@pytest.mark.parametrize("mode", [0, 1])
def test_subpixel_matches_oracle(ctx, mode):
    import shadernn_amd as snn

    x = _rand((2, 9, 7, 4), 31)
    plan = snn.subpixel_plan(ctx, 2, 9, 7, 4, 2, mode)
    y = plan(snn.Tensor.from_numpy(ctx, x)).numpy()
    np.testing.assert_allclose(y, O.subpixel(x, 2, mode), **TOL)


def test_c4hw4_edge_conversion_roundtrip(ctx):
    import shadernn_amd as snn

    x = _rand((1, 5, 6, 7), 41)
    t = snn.Tensor.from_numpy(ctx, x)
    c4 = t.numpy_c4hw4()
    assert c4.shape == (1, 2, 5, 6, 4)
    np.testing.assert_array_equal(c4[0, 1, :, :, 3], 0)  # pad channel
    np.testing.assert_array_equal(c4[0, 0, :, :, 2], x[0, :, :, 2])
    t2 = snn.Tensor(ctx, 1, 5, 6, 7)
    t2.upload_c4hw4(c4)
    np.testing.assert_array_equal(t2.numpy(), x)


def test_bad_arguments_fail_loudly(ctx):
    import shadernn_amd as snn

    w = _rand((4, 3, 3, 3), 1)
    plan = snn.conv2d_plan(ctx, 1, 8, 8, w)
    bad = snn.Tensor(ctx, 1, 8, 8, 5)
    out = snn.Tensor(ctx, 1, 8, 8, 4)
    with pytest.raises(snn.SnnHipError):
        plan.run(bad, out)


def test_graph_capture_replays_a_layer_sequence(ctx):
    """snnhip_graph_*: plans run between begin/end are recorded, one launch replays them (the command-buffer replay of the reference)."""
    import shadernn_amd as snn
    from shadernn_amd import models

    net = models.resnet18(seed=5, num_classes=10, width=8)
    x = np.random.default_rng(1).random((2, 64, 64, 3), dtype=np.float32)
    r = snn.GraphRunner(ctx, net, 2, 64, 64)
    want = r(x)
    with snn.Graph.capture(ctx) as g:
        r.run_device()
    assert g.num_nodes() >= len(r.steps)
    r.y.fill(0.0)
    g.launch()
    g.launch()
    np.testing.assert_array_equal(r.y.numpy(), want)
    x2 = np.random.default_rng(2).random((2, 64, 64, 3), dtype=np.float32)
    r.x.upload(x2)  # same tensors, new contents
    g.launch()
    np.testing.assert_allclose(r.y.numpy(), O.forward(net, x2), **TOL)
    g.destroy()
