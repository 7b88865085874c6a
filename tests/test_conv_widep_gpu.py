"""conv2d_widep_f16.hip: the persistent form of the fp16 3x3 stride-1 128 -> 128 kernel (hand-counted vmcnt, buffer-descriptor DMA and stores, work
interleaved into the K-steps).  It is the default for these layers, so tests/test_conv_wide_gpu.py already runs it against the oracle wherever a
case has IC = OC = 128; this file pins what is specific to it: bit-identity with the block-per-tile kernel (SNNHIP_WIDE_PERSIST=0) over ragged maps,
padding modes and epilogues; grids of 1, 3 and 7 resident blocks (SNNHIP_WIDEP_GRID), whose contiguous tile runs cross image boundaries in the
middle of a run -- with the statistics records of chain rule F and the normalising staging of graph rule I switched on; and that it declines what
it does not implement.  Semantics: shadertemplate_vk_conv2d.comp:148-347."""
import numpy as np
import pytest

import oracle_lib as O
from test_conv_wide_gpu import TOLH, _run
from test_ops_gpu import _bn, _rand

pytestmark = pytest.mark.gpu

# N, H, W: one tile; ragged in both directions; several images; a map narrower than a tile; a long row of tiles
MAPS = [(1, 8, 32), (2, 19, 45), (1, 37, 70), (3, 9, 33), (4, 5, 7), (1, 8, 300), (5, 17, 65)]


@pytest.mark.parametrize("grid", ["", "1", "3", "7", "8", "16"], ids=lambda g: "grid" + (g or "auto"))  # (8, 16: the XCD-segment run layout)
@pytest.mark.parametrize("nhw", MAPS, ids=lambda c: "x".join(map(str, c)))
def test_widep_is_bit_identical_to_the_block_per_tile_kernel(ctx, monkeypatch, nhw, grid):
    N, H, W = nhw
    x = _rand((N, H, W, 128), 181)
    w = _rand((128, 128, 3, 3), 182, 1.0 / np.sqrt(128 * 9))
    b = _rand((128,), 183, 0.1)
    bn = _bn(128, 184)
    pads = O.padding_offsets("same", 3)
    monkeypatch.setenv("SNNHIP_CONV", "wide")
    for pad_mode, act, use_bn in (("constant", "relu", False), ("constant", "", False), ("reflect", "relu", True), ("replicate", "", True)):
        if pad_mode == "reflect" and min(H, W) < 2:
            continue
        monkeypatch.delenv("SNNHIP_WIDE_PERSIST", raising=False)
        if grid:
            monkeypatch.setenv("SNNHIP_WIDEP_GRID", grid)
        y, desc = _run(ctx, x, w, b, pad_mode, act, bn if use_bn else None)
        assert "persistent" in desc, desc
        if grid:
            tiles = N * ((H + 7) // 8) * ((W + 31) // 32)
            assert "blocks=%d " % min(int(grid), tiles) in desc, desc
        monkeypatch.setenv("SNNHIP_WIDE_PERSIST", "0")
        y0, desc0 = _run(ctx, x, w, b, pad_mode, act, bn if use_bn else None)
        assert "persistent" not in desc0 and "wide" in desc0, desc0
        np.testing.assert_array_equal(y, y0, err_msg=desc + " vs " + desc0)
        if not grid:
            want = O._h(O.conv2d(O._h(x), O._h(w), b, 1, pads, pad_mode, act, 0.0, bn if use_bn else None))
            np.testing.assert_allclose(y, want, err_msg=desc, **TOLH)


@pytest.mark.parametrize("grid", ["", "1", "3", "7", "8", "16"], ids=lambda g: "grid" + (g or "auto"))  # (8, 16: the XCD-segment run layout)
@pytest.mark.parametrize("n,h,w,offset", [(3, 19, 45, 0.0), (5, 9, 70, 6.0), (2, 30, 33, 0.0)])
def test_widep_block_records_feed_the_instancenorm_for_any_grid(ctx, monkeypatch, n, h, w, offset, grid):
    """Chain rule F on the persistent kernel: ONE record per (block, image), written when a block's run of tiles leaves the image; the block that
    writes an image's last record folds it.  With 1, 3 or 7 blocks the runs end in the middle of images, whole images fall inside one run, and a
    single block folds every image itself.  Against the separate launches (same stored tensor, the tight comparison) and the block-per-tile
    kernel's per-tile records; twice through the same plan with different data (counters back at zero, no stale record)."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_CONV", "wide")
    monkeypatch.setenv("SNNHIP_NORM_FUSION_MIN_MB", "0")
    if grid:
        monkeypatch.setenv("SNNHIP_WIDEP_GRID", grid)
    x, wt = _rand((n, h, w, 128), 1), _rand((128, 128, 3, 3), 2, 1.0 / np.sqrt(128 * 9))
    b = _rand((128,), 3, 0.5) + offset
    beta, gamma = _rand((128,), 4, 0.3), 1.0 + _rand((128,), 5, 0.2)
    pad = snn.pad_plan(ctx, n, h, w, 128, (1, 1, 1, 1), "reflect")
    conv = snn.conv2d_plan(ctx, n, h + 2, w + 2, wt, b, stride=1, pads=(0, 0, 0, 0), act="", dtype=snn.F16)
    oh, ow = conv.out_shape()[1:3]
    norm = snn.instancenorm_plan(ctx, n, oh, ow, 128, beta, gamma, act="relu")
    fused = snn.chain_plan(ctx, [pad, conv, norm])
    d = fused.describe()
    assert fused.num_steps() == 1 and "persistent" in d and "+tile-stats+fold" in d, d
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    y = fused(xt).numpy()
    two = norm(conv(pad(xt))).numpy()
    np.testing.assert_allclose(y, two, rtol=2e-3, atol=2e-3, err_msg=d)
    x2 = (0.5 * _rand((n, h, w, 128), 11) + 0.75).astype(np.float32)
    xt2 = snn.Tensor.from_numpy(ctx, x2, dtype=snn.F16)
    y2 = fused(xt2).numpy()
    np.testing.assert_allclose(y2, norm(conv(pad(xt2))).numpy(), rtol=2e-3, atol=2e-3, err_msg="second input through the same plan: " + d)
    np.testing.assert_array_equal(fused(xt).numpy(), y)
    monkeypatch.setenv("SNNHIP_WIDE_PERSIST", "0")
    old = snn.chain_plan(ctx, [pad, conv, norm])
    assert "persistent" not in old.describe() and "+tile-stats" in old.describe(), old.describe()
    np.testing.assert_allclose(y, old(xt).numpy(), rtol=2e-3, atol=2e-3, err_msg=d + " vs " + old.describe())


@pytest.mark.parametrize("grid", ["", "3"], ids=lambda g: "grid" + (g or "auto"))
@pytest.mark.parametrize("act", ["relu", ""])
def test_widep_normalises_what_it_stages_like_the_block_per_tile_kernel(ctx, monkeypatch, act, grid):
    """Graph rule I: InstanceNorm [-> ReLU] -> reflect Pad -> Conv2D with the norm applied in LDS behind the DMA, inside the K-steps of the chunk that
    copied the rows.  Same result as the block-per-tile kernel's LDS pass and as the separate launches; a norm activation the persistent kernel does
    not implement goes back to conv2d_wide_kernel."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_CONV", "wide")
    if grid:
        monkeypatch.setenv("SNNHIP_WIDEP_GRID", grid)
    n, h, w = 3, 21, 70
    x, wt, b = _rand((n, h, w, 128), 21) * 2.0 + 0.5, _rand((128, 128, 3, 3), 22, 1.0 / np.sqrt(128 * 9)), _rand((128,), 23, 0.2)
    beta, gamma = _rand((128,), 24, 0.3), 1.0 + _rand((128,), 25, 0.2)
    norm = snn.instancenorm_plan(ctx, n, h, w, 128, beta, gamma, act=act)
    pad = snn.pad_plan(ctx, n, h, w, 128, (1, 1, 1, 1), "reflect")
    conv = snn.conv2d_plan(ctx, n, h + 2, w + 2, wt, b, stride=1, pads=(0, 0, 0, 0), act="relu", dtype=snn.F16)
    fused = snn.chain_plan(ctx, [norm, pad, conv])
    d = fused.describe()
    assert fused.num_steps() == 1 and "in LDS behind the DMA) -> conv2d_mfma_wide_f16" in d and "persistent" in d, d
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    y = fused(xt).numpy()
    np.testing.assert_array_equal(y, conv(pad(norm(xt))).numpy(), err_msg=d)  # (same arithmetic and rounding point as the normalise sweep)
    monkeypatch.setenv("SNNHIP_WIDE_PERSIST", "0")
    old = snn.chain_plan(ctx, [norm, pad, conv])
    assert "persistent" not in old.describe(), old.describe()
    np.testing.assert_array_equal(y, old(xt).numpy(), err_msg=d + " vs " + old.describe())
    monkeypatch.delenv("SNNHIP_WIDE_PERSIST")
    leaky = snn.chain_plan(ctx, [snn.instancenorm_plan(ctx, n, h, w, 128, beta, gamma, act="leakyRelu", leaky=0.1), pad, conv])
    assert "persistent" not in leaky.describe() and "wide" in leaky.describe(), leaky.describe()


def test_widep_declines_what_it_does_not_implement(ctx, monkeypatch):
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_CONV", "wide")
    b = _rand((128,), 3, 0.1)
    for ic, oc, act in ((64, 128, "relu"), (128, 64, "relu"), (128, 128, "tanh")):
        w = _rand((oc, ic, 3, 3), 2, 0.05)
        plan = snn.conv2d_plan(ctx, 1, 16, 64, w, b[:oc], stride=1, pads=O.padding_offsets("same", 3), act=act, dtype=snn.F16)
        assert "persistent" not in plan.describe() and "wide" in plan.describe(), plan.describe()
        plan.destroy()
