"""ESPCN 2x end-to-end (BASELINE config 2): per-layer plans and the fused chain vs the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,h,w", [(1, 32, 32), (1, 72, 96), (2, 19, 71), (1, 8, 64), (1, 9, 130), (3, 5, 7)])
@pytest.mark.parametrize("fused", [False, True])
def test_espcn_matches_oracle(ctx, n, h, w, fused):
    import shadernn_amd as snn
    from shadernn_amd import models

    net = models.espcn_weights(seed=1)
    x = np.random.default_rng(7767517).random((n, h, w, 1), dtype=np.float32)
    runner = snn.EspcnRunner(ctx, net, n, h, w, fused=fused)
    y = runner(x)
    want, layers = O.forward(net, x, return_layers=True)
    assert y.shape == (n, 2 * h, 2 * w, 1)
    np.testing.assert_allclose(y, want, err_msg="; ".join(runner.describe()), **TOL)
    if not fused:  # layer-by-layer comparison in the style of the reference's model tests (resnet18Test.cpp:84-140)
        for got, exp in zip(runner.layer_outputs(), layers):
            np.testing.assert_allclose(got, exp, **TOL)
    else:
        assert "fused[conv5x5" in runner.describe()[0] and "depth_to_space" in runner.describe()[0] and " -> " in runner.describe()[0]


@pytest.mark.parametrize("n,h,w", [(1, 32, 32), (2, 19, 71), (1, 40, 130)])
def test_espcn_stream_fusion_matches_oracle(ctx, n, h, w, monkeypatch):
    """The single row-streaming kernel (rule C) is opt-in via SNNHIP_ESPCN_FUSION=stream and must stay parity-green."""
    import shadernn_amd as snn
    from shadernn_amd import models

    monkeypatch.setenv("SNNHIP_ESPCN_FUSION", "stream")
    net = models.espcn_weights(seed=2)
    x = np.random.default_rng(3).random((n, h, w, 1), dtype=np.float32)
    runner = snn.EspcnRunner(ctx, net, n, h, w, fused=True)
    assert "stream" in runner.describe()[0]
    np.testing.assert_allclose(runner(x), O.forward(net, x), **TOL)


def test_espcn_stream_full_size(ctx, monkeypatch):
    import shadernn_amd as snn
    from shadernn_amd import models

    net = models.espcn_weights(seed=1)
    x = np.random.default_rng(1).random((1, 1080, 1920, 1), dtype=np.float32)
    y_pair = snn.EspcnRunner(ctx, net, 1, 1080, 1920, fused=True)(x)
    monkeypatch.setenv("SNNHIP_ESPCN_FUSION", "stream")
    y_stream = snn.EspcnRunner(ctx, net, 1, 1080, 1920, fused=True)(x)
    np.testing.assert_allclose(y_stream, y_pair, rtol=1e-5, atol=1e-5)


def test_espcn_fused_with_bn_and_other_activations(ctx):
    """The fusion rules carry the full epilogue (bias, BN, any plain activation), not just ESPCN's relu."""
    import shadernn_amd as snn
    from shadernn_amd import models

    net = models.espcn_weights(seed=3)
    rng = np.random.default_rng(5)
    for i, act in enumerate(["leakyRelu", "sigmoid", "tanh"]):
        l = net["layers"][i]
        l["activation"] = act
        l["alpha"] = 0.2
        c = l["oc"]
        l["bn"] = {"beta": rng.uniform(-0.1, 0.1, c).astype(np.float32), "gamma": rng.uniform(0.5, 1.5, c).astype(np.float32),
                   "mean": rng.uniform(-0.1, 0.1, c).astype(np.float32), "var": rng.uniform(0.5, 1.5, c).astype(np.float32)}
    x = rng.random((1, 21, 67, 1), dtype=np.float32)
    runner = snn.EspcnRunner(ctx, net, 1, 21, 67, fused=True)
    np.testing.assert_allclose(runner(x), O.forward(net, x), **TOL)


def test_espcn_full_size_properties(ctx):
    """1080p (the benchmark size): fused and per-layer paths agree everywhere; a window of the image equals the
    oracle run on that window + halo (translation equivariance away from the border)."""
    import shadernn_amd as snn
    from shadernn_amd import models

    net = models.espcn_weights(seed=1)
    H, W = 1080, 1920
    x = np.random.default_rng(1).random((1, H, W, 1), dtype=np.float32)
    y_f = snn.EspcnRunner(ctx, net, 1, H, W, fused=True)(x)
    y_u = snn.EspcnRunner(ctx, net, 1, H, W, fused=False)(x)
    np.testing.assert_allclose(y_f, y_u, rtol=1e-5, atol=1e-5)
    assert np.isfinite(y_f).all() and np.abs(y_f).max() <= 1.0
    # oracle on crops: top-left corner (true borders) and an interior window (halo 4 discarded)
    crop = O.forward(net, x[:, :40, :48, :])
    np.testing.assert_allclose(y_f[:, : 2 * 36, : 2 * 44, :], crop[:, : 2 * 36, : 2 * 44, :], **TOL)
    y0, x0 = 500, 900
    crop = O.forward(net, x[:, y0 - 4 : y0 + 36, x0 - 4 : x0 + 44, :])
    np.testing.assert_allclose(y_f[:, 2 * y0 : 2 * (y0 + 32), 2 * x0 : 2 * (x0 + 40), :], crop[:, 8 : 8 + 64, 8 : 8 + 80, :], **TOL)
    crop = O.forward(net, x[:, H - 40 :, W - 48 :, :])  # bottom-right corner
    np.testing.assert_allclose(y_f[:, 2 * (H - 36) :, 2 * (W - 44) :, :], crop[:, 8:, 8:, :], **TOL)


def test_chain_rejects_unfusable(ctx):
    import shadernn_amd as snn

    w = np.zeros((8, 3, 3, 3), np.float32)
    p = snn.conv2d_plan(ctx, 1, 8, 8, w)
    with pytest.raises(snn.SnnHipError) as e:
        snn.chain_plan(ctx, [p])
    assert e.value.code == -3


VARIANTS = [("direct", "direct"), ("wino", "direct"), ("wino", "wino"), ("direct", "wino")]


@pytest.mark.parametrize("a_mode,b_mode", VARIANTS)
@pytest.mark.parametrize("n,h,w", [(1, 72, 96), (2, 19, 71), (1, 33, 130), (1, 16, 32)])
def test_espcn_kernel_variants_match_oracle(ctx, monkeypatch, a_mode, b_mode, n, h, w):
    """Every selectable kernel of the fused chain (SNNHIP_ESPCN_A = wino | direct, SNNHIP_ESPCN_B = direct | wino) against the oracle: the
    Winograd F(2x2,3x3) evaluations stay inside the same 1e-4 bound."""
    import shadernn_amd as snn
    from shadernn_amd import models

    monkeypatch.setenv("SNNHIP_ESPCN_A", a_mode)
    monkeypatch.setenv("SNNHIP_ESPCN_B", b_mode)
    net = models.espcn_weights(seed=4)
    x = np.random.default_rng(11).random((n, h, w, 1), dtype=np.float32)
    runner = snn.EspcnRunner(ctx, net, n, h, w, fused=True)
    desc = runner.describe()[0]
    assert ("winograd" in desc.split(" -> ")[0]) == (a_mode == "wino"), desc
    assert ("mfma_f32_4x4x1" in desc) == b_mode.startswith("wino"), desc
    np.testing.assert_allclose(runner(x), O.forward(net, x), err_msg=desc, **TOL)


@pytest.mark.parametrize("a_mode,b_mode", [("direct", "direct"), ("wino", "wino")])
def test_espcn_kernel_variants_full_size(ctx, monkeypatch, a_mode, b_mode):
    """1080p: the persistent tile loops (4080 / 2040 tiles over 512 resident blocks) agree with the per-layer path."""
    import shadernn_amd as snn
    from shadernn_amd import models

    net = models.espcn_weights(seed=1)
    H, W = 1080, 1920
    x = np.random.default_rng(2).random((1, H, W, 1), dtype=np.float32)
    y_u = snn.EspcnRunner(ctx, net, 1, H, W, fused=False)(x)
    monkeypatch.setenv("SNNHIP_ESPCN_A", a_mode)
    monkeypatch.setenv("SNNHIP_ESPCN_B", b_mode)
    y_f = snn.EspcnRunner(ctx, net, 1, H, W, fused=True)(x)
    np.testing.assert_allclose(y_f, y_u, rtol=2e-5, atol=2e-5)
