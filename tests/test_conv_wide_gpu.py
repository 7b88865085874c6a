"""conv2d_wide_f16.hip: the fp16 3x3 stride-1 kernel with 4 x NT MFMA register blocks (256 / 512-pixel tiles), forced with SNNHIP_CONV=wide on
shapes small enough for the oracle: ragged tile edges in both directions, every block shape (128 / 64 / 32 output channels per block, 16- and
32-channel chunks), padding modes, epilogues, the fused Pad / UpSampling staging path and the fused residual Add -- against the quantised
oracle and against the 128-pixel kernel (SNNHIP_CONV=mfma) on the same inputs."""
import numpy as np
import pytest

import oracle_lib as O
from test_ops_gpu import _bn, _rand

pytestmark = pytest.mark.gpu
TOLH = dict(rtol=4e-3, atol=4e-3)

# N, H, W, IC, OC
SHAPES = [(1, 16, 64, 32, 128), (2, 19, 45, 64, 128), (1, 9, 33, 16, 128), (1, 37, 70, 128, 128), (2, 18, 40, 32, 64), (1, 33, 35, 48, 64), (1, 20, 66, 64, 32),
          (3, 7, 9, 16, 96), (1, 8, 32, 256, 256)]


def _run(ctx, x, w, b, pad_mode, act, bn, pads=None):
    import shadernn_amd as snn

    n, h, ww, _ = x.shape
    plan = snn.conv2d_plan(ctx, n, h, ww, w, b, stride=1, pads=pads or O.padding_offsets("same", 3), pad_mode=pad_mode, act=act, bn=bn, dtype=snn.F16)
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    y, desc = plan(xt).numpy(), plan.describe()
    plan.destroy()
    return y, desc


@pytest.mark.parametrize("shape", SHAPES, ids=lambda c: "x".join(map(str, c)))
def test_wide_matches_quantised_oracle_and_narrow_kernel(ctx, monkeypatch, shape):
    N, H, W, IC, OC = shape
    x = _rand((N, H, W, IC), 81)
    w = _rand((OC, IC, 3, 3), 82, 1.0 / np.sqrt(IC * 9))
    b = _rand((OC,), 83, 0.1)
    bn = _bn(OC, 84)
    pads = O.padding_offsets("same", 3)
    for pad_mode, act, use_bn in (("constant", "relu", True), ("reflect", "tanh", False), ("replicate", "", True)):
        monkeypatch.setenv("SNNHIP_CONV", "wide")
        y, desc = _run(ctx, x, w, b, pad_mode, act, bn if use_bn else None)
        assert "conv2d_mfma_wide_f16" in desc, desc
        want = O._h(O.conv2d(O._h(x), O._h(w), b, 1, pads, pad_mode, act, 0.0, bn if use_bn else None))
        np.testing.assert_allclose(y, want, err_msg=desc, **TOLH)
        monkeypatch.setenv("SNNHIP_CONV", "mfma")
        y2, desc2 = _run(ctx, x, w, b, pad_mode, act, bn if use_bn else None)
        assert "wide" not in desc2, desc2
        np.testing.assert_allclose(y, y2, err_msg=desc + " vs " + desc2, rtol=2e-3, atol=2e-3)


def test_wide_valid_padding_and_block_widths(ctx, monkeypatch):
    """'valid' convolutions (no border offsets; under the reference's size rule the output keeps the input extent and the taps past the bottom /
    right edge read zeros) and the forced block widths of a 128-channel layer."""
    x = _rand((1, 21, 50, 32), 5)
    w = _rand((128, 32, 3, 3), 6, 1.0 / np.sqrt(32 * 9))
    b = _rand((128,), 7, 0.1)
    want = O._h(O.conv2d(O._h(x), O._h(w), b, 1, (0, 0, 0, 0), "constant", "relu", 0.0, None))
    monkeypatch.setenv("SNNHIP_CONV", "wide")
    for bn_w in ("128", "64", "32"):
        monkeypatch.setenv("SNNHIP_WIDE_BN", bn_w)
        y, desc = _run(ctx, x, w, b, "constant", "relu", None, pads=(0, 0, 0, 0))
        assert "x %soc" % bn_w in desc, desc
        assert y.shape == want.shape == (1, 21, 50, 128)
        np.testing.assert_allclose(y, want, err_msg=desc, **TOLH)


@pytest.mark.parametrize("with_up", [False, True])
@pytest.mark.parametrize("oc", [128, 64, 32])
def test_wide_with_fused_pad_upsample_and_add(ctx, monkeypatch, oc, with_up):
    """The chain planner's fusions land on the wide kernel too: [UpSampling ->] reflect Pad -> Conv2D (rule D) and Conv2D -> Add (rule E)."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_CONV", "wide")
    n, h, w_, ic = 2, 13, 21, 32
    x, wt, b = _rand((n, h, w_, ic), 1), _rand((oc, ic, 3, 3), 2, 1.0 / np.sqrt(ic * 9)), _rand((oc,), 3, 0.2)
    plans, hh, ww = [], h, w_
    if with_up:
        plans.append(snn.upsample_plan(ctx, n, h, w_, ic, 2.0, "nearest"))
        hh, ww = 2 * h, 2 * w_
    plans.append(snn.pad_plan(ctx, n, hh, ww, ic, (1, 1, 1, 1), "reflect"))
    plans.append(snn.conv2d_plan(ctx, n, hh + 2, ww + 2, wt, b, stride=1, pads=(0, 0, 0, 0), act="relu", dtype=snn.F16))
    chain = snn.chain_plan(ctx, plans)
    assert chain.num_steps() == 1 and "wide" in chain.describe() and "+pad(reflect)" in chain.describe(), chain.describe()
    assert ("+upsample" in chain.describe()) == with_up
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    y = chain(xt).numpy()
    t = O._h(x)
    if with_up:
        t = O.upsample(t, 2.0, "nearest")
    t = O.pad(t, (1, 1, 1, 1), "reflect")
    want = O._h(O.conv2d(t, O._h(wt), b, 1, (0, 0, 0, 0), "constant", "relu", 0.0, None))
    np.testing.assert_allclose(y, want, err_msg=chain.describe(), **TOLH)
    # conv -> add
    conv = snn.conv2d_plan(ctx, n, hh, ww, _rand((oc, oc, 3, 3), 4, 1.0 / np.sqrt(oc * 9)), _rand((oc,), 5, 0.2), stride=1, pads=(1, 1, 1, 1), act="", dtype=snn.F16) if oc % 16 == 0 else None
    add = snn.add_plan(ctx, n, hh, ww, oc, act="relu")
    fused = snn.chain_plan(ctx, [conv, add])
    assert fused.num_steps() == 1 and "wide" in fused.describe() and "+add" in fused.describe(), fused.describe()
    a, r = _rand((n, hh, ww, oc), 6), _rand((n, hh, ww, oc), 7)
    at, rt = snn.Tensor.from_numpy(ctx, a, dtype=snn.F16), snn.Tensor.from_numpy(ctx, r, dtype=snn.F16)
    got = fused([at, rt]).numpy()
    two = add([conv(at), rt]).numpy()
    np.testing.assert_array_equal(got, two)


def test_wide_is_the_default_on_large_maps_only(ctx):
    import shadernn_amd as snn

    w = _rand((128, 128, 3, 3), 1, 0.03)
    big = snn.conv2d_plan(ctx, 8, 180, 320, w, None, stride=1, pads=(1, 1, 1, 1), dtype=snn.F16)
    small = snn.conv2d_plan(ctx, 1, 56, 56, w, None, stride=1, pads=(1, 1, 1, 1), dtype=snn.F16)
    one = snn.conv2d_plan(ctx, 1, 183, 323, w, None, stride=1, pads=(1, 1, 1, 1), dtype=snn.F16)
    assert "wide" in one.describe(), one.describe()
    s2 = snn.conv2d_plan(ctx, 8, 180, 320, w, None, stride=2, pads=(1, 1, 1, 1), dtype=snn.F16)
    f32 = snn.conv2d_plan(ctx, 8, 180, 320, w, None, stride=1, pads=(1, 1, 1, 1))
    assert "wide" in big.describe(), big.describe()
    for p in (small, s2, f32):
        assert "wide" not in p.describe(), p.describe()


def test_wide_full_size_layer_properties(ctx, monkeypatch):
    """Candy's residual-block layer at its benchmark size (16 x 183 x 323 x 128 -> 128): translation equivariance against the oracle on crops
    (corner + interior), batch-index property, agreement with the 128-pixel kernel everywhere."""
    import shadernn_amd as snn

    N, H, W, C = 4, 183, 323, 128
    x1 = _rand((1, H, W, C), 11)
    x = np.repeat(x1, N, axis=0)
    w = _rand((C, C, 3, 3), 12, 1.0 / np.sqrt(C * 9))
    b = _rand((C,), 13, 0.1)
    y, desc = _run(ctx, x, w, b, "constant", "relu", None)
    assert "wide" in desc, desc
    for i in range(1, N):
        np.testing.assert_array_equal(y[i], y[0])
    monkeypatch.setenv("SNNHIP_CONV", "mfma")
    y2, desc2 = _run(ctx, x1, w, b, "constant", "relu", None)
    np.testing.assert_allclose(y[:1], y2, err_msg=desc + " vs " + desc2, rtol=2e-3, atol=2e-3)
    pads = O.padding_offsets("same", 3)
    crop = O._h(O.conv2d(O._h(x1[:, :20, :40]), O._h(w), b, 1, pads, "constant", "relu", 0.0, None))
    np.testing.assert_allclose(y[:1, :19, :39], crop[:, :19, :39], **TOLH)
    crop = O._h(O.conv2d(O._h(x1[:, H - 20 :, W - 40 :]), O._h(w), b, 1, pads, "constant", "relu", 0.0, None))
    np.testing.assert_allclose(y[:1, H - 19 :, W - 39 :], crop[:, 1:, 1:], **TOLH)
    crop = O._h(O.conv2d(O._h(x1[:, 90:120, 150:200]), O._h(w), b, 1, pads, "constant", "relu", 0.0, None))
    np.testing.assert_allclose(y[:1, 91:119, 151:199], crop[:, 1:-1, 1:-1], **TOLH)


# ---- conv2d_stem_f16.hip: the fp16 9x9 RGB stem (IC <= 4, 4 taps x 4 channels per K step, weights in registers)
STEM = [(1, 40, 50, 3, 32), (2, 33, 31, 3, 32), (1, 9, 70, 1, 32), (1, 64, 64, 4, 64), (3, 5, 7, 3, 32), (1, 100, 37, 2, 96)]


@pytest.mark.parametrize("shape", STEM, ids=lambda c: "x".join(map(str, c)))
def test_stem_matches_quantised_oracle_and_tap_pair_kernel(ctx, monkeypatch, shape):
    N, H, W, IC, OC = shape
    x = _rand((N, H, W, IC), 31)
    w = _rand((OC, IC, 9, 9), 32, 1.0 / np.sqrt(IC * 81))
    b = _rand((OC,), 33, 0.1)
    bn = _bn(OC, 34)
    pads = O.padding_offsets("same", 9)

    def run(pad_mode, act, bnp):
        import shadernn_amd as snn

        plan = snn.conv2d_plan(ctx, N, H, W, w, b, stride=1, pads=pads, pad_mode=pad_mode, act=act, bn=bnp, dtype=snn.F16)
        y, d = plan(snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)).numpy(), plan.describe()
        plan.destroy()
        return y, d

    for pad_mode, act, use_bn in (("constant", "relu", False), ("reflect", "tanh", True), ("replicate", "", False)):
        if pad_mode == "reflect" and min(H, W) < 5:
            continue  # a reflection needs the image to be wider than the border
        monkeypatch.delenv("SNNHIP_CONV_STEM", raising=False)
        y, desc = run(pad_mode, act, bn if use_bn else None)
        assert "conv2d_mfma_stem_f16" in desc, desc
        want = O._h(O.conv2d(O._h(x), O._h(w), b, 1, pads, pad_mode, act, 0.0, bn if use_bn else None))
        np.testing.assert_allclose(y, want, err_msg=desc, **TOLH)
        monkeypatch.setenv("SNNHIP_CONV_STEM", "0")
        y2, desc2 = run(pad_mode, act, bn if use_bn else None)
        assert "stem" not in desc2, desc2
        np.testing.assert_allclose(y, y2, err_msg=desc + " vs " + desc2, rtol=2e-3, atol=2e-3)


def test_stem_with_fused_reflect_pad_and_non_finite_neighbours(ctx):
    """Candy's first layers: reflect Pad(4) -> 9x9 'valid' convolution as one launch (chain rule D).  An inf pixel reaches exactly the outputs whose
    receptive field holds it (the zero-weight operand slots of the kernel's last K step must not turn it into NaN elsewhere)."""
    import shadernn_amd as snn

    n, h, w_, ic, oc = 1, 45, 52, 3, 32
    x, wt, b = _rand((n, h, w_, ic), 1), _rand((oc, ic, 9, 9), 2, 1.0 / np.sqrt(ic * 81)), _rand((oc,), 3, 0.2)
    pad = snn.pad_plan(ctx, n, h, w_, ic, (4, 4, 4, 4), "reflect")
    conv = snn.conv2d_plan(ctx, n, h + 8, w_ + 8, wt, b, stride=1, pads=(0, 0, 0, 0), act="relu", dtype=snn.F16)
    chain = snn.chain_plan(ctx, [pad, conv])
    assert chain.num_steps() == 1 and "stem" in chain.describe() and "+pad(reflect)" in chain.describe(), chain.describe()
    y = chain(snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)).numpy()
    want = O._h(O.conv2d(O.pad(O._h(x), (4, 4, 4, 4), "reflect"), O._h(wt), b, 1, (0, 0, 0, 0), "constant", "relu", 0.0, None))
    assert y.shape == want.shape
    np.testing.assert_allclose(y, want, err_msg=chain.describe(), **TOLH)
    xi = x.copy()
    xi[0, 20, 30, 1] = np.inf
    wpos = np.abs(wt) + 0.01  # every tap weight non-zero and positive: inf stays +inf inside the field
    conv2 = snn.conv2d_plan(ctx, n, h, w_, wpos, b, stride=1, pads=O.padding_offsets("same", 9), act="", dtype=snn.F16)
    assert "stem" in conv2.describe()
    yi = conv2(snn.Tensor.from_numpy(ctx, xi, dtype=snn.F16)).numpy()
    bad = ~np.isfinite(yi[0, :, :, 0])
    field = np.zeros_like(bad)
    field[20 - 4 : 20 + 5, 30 - 4 : 30 + 5] = True
    assert (bad == field).all() and not np.isnan(yi).any()


# ---- conv2d_stem_f32.hip: the fp32 RGB stems (IC <= 4; 3x3 stride 1 / 2, 7x7 stride 2), 2 MFMAs per tap, weights in registers
STEM32 = [(1, 224, 224, 3, 64, 3, 1), (2, 64, 64, 3, 32, 3, 2), (2, 75, 61, 3, 64, 7, 2), (2, 224, 224, 3, 64, 7, 2), (1, 33, 47, 1, 32, 3, 1), (1, 40, 40, 4, 96, 7, 2), (3, 9, 11, 3, 32, 3, 2),
          (1, 17, 130, 2, 64, 3, 1)]


@pytest.mark.parametrize("rows", [4, 2], ids=["16-row-tiles", "8-row-tiles"])
@pytest.mark.parametrize("shape", STEM32, ids=lambda c: "x".join(map(str, c)))
def test_stem32_matches_oracle_and_tap_pair_kernel(ctx, monkeypatch, shape, rows):
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_STEM_ROWS", str(rows))  # (default: 8-row tiles on grids below two blocks per CU, 16-row tiles otherwise)
    N, H, W, IC, OC, k, s = shape
    x = _rand((N, H, W, IC), 41)
    w = _rand((OC, IC, k, k), 42, 1.0 / np.sqrt(IC * k * k))
    b = _rand((OC,), 43, 0.1)
    bn = _bn(OC, 44)
    pads = O.padding_offsets("same", k)
    tol = dict(rtol=1e-4, atol=1e-4)

    def run(pad_mode, act, bnp):
        plan = snn.conv2d_plan(ctx, N, H, W, w, b, stride=s, pads=pads, pad_mode=pad_mode, act=act, bn=bnp)
        y, d = plan(snn.Tensor.from_numpy(ctx, x)).numpy(), plan.describe()
        plan.destroy()
        return y, d

    for pad_mode, act, use_bn in (("constant", "relu", True), ("reflect", "tanh", False), ("replicate", "leakyRelu", True)):
        if pad_mode == "reflect" and min(H, W) <= k // 2:
            continue
        monkeypatch.delenv("SNNHIP_CONV_STEM", raising=False)
        y, desc = run(pad_mode, act, bn if use_bn else None)
        assert "conv2d_mfma_stem_f32" in desc and "tile=%dx32px" % (4 * rows) in desc, desc
        want = O.conv2d(x, w, b, s, pads, pad_mode, act, 0.0, bn if use_bn else None)
        assert y.shape == want.shape
        np.testing.assert_allclose(y, want, err_msg=desc, **tol)
        if k == 7 and IC == 3:  # the RGB 7x7 stride-2 stem packs its reduction densely (74 K steps); SNNHIP_STEM_DENSE=0 gives the general form back
            assert "dense (tap, channel) K" in desc, desc
            monkeypatch.setenv("SNNHIP_STEM_DENSE", "0")
            y3, desc3 = run(pad_mode, act, bn if use_bn else None)
            monkeypatch.delenv("SNNHIP_STEM_DENSE")
            assert "2 MFMAs per tap" in desc3, desc3
            np.testing.assert_allclose(y, y3, err_msg=desc + " vs " + desc3, rtol=2e-5, atol=2e-5)
        else:
            assert "2 MFMAs per tap" in desc, desc
        monkeypatch.setenv("SNNHIP_CONV_STEM", "0")
        y2, desc2 = run(pad_mode, act, bn if use_bn else None)
        assert "stem" not in desc2, desc2
        np.testing.assert_allclose(y, y2, err_msg=desc + " vs " + desc2, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("shape", [(2, 224, 224, 64, "relu", True), (1, 75, 61, 32, "tanh", False), (3, 131, 58, 96, "leakyRelu", True), (1, 33, 230, 64, "", True)],
                         ids=["resnet_head", "ragged_tanh", "three_oc_blocks", "wide_short"])
def test_stem_maxpool_runs_as_one_launch(ctx, monkeypatch, shape):
    """Chain rule J: Conv2D 7x7 stride 2 of an RGB image -> MaxPooling2D 3x3 stride 2 (the head of ResNet-18) as one launch: the pooling runs in the dense
    stem kernel's epilogue (windows clipped at the bottom / right, no padding on top / left, maximum floored at -100000.0 like the separate layer).
    Bit-identical to the two separate launches (the same convolution arithmetic, then exact maxima), within 1e-4 of the oracle; the switch turns
    the rule off."""
    import shadernn_amd as snn

    N, H, W, OC, act, use_bn = shape
    x = _rand((N, H, W, 3), 91)
    w = _rand((OC, 3, 7, 7), 92, 1.0 / np.sqrt(147))
    b = _rand((OC,), 93, 0.1)
    bn = _bn(OC, 94) if use_bn else None
    pads = O.padding_offsets("same", 7)
    conv = snn.conv2d_plan(ctx, N, H, W, w, b, stride=2, pads=pads, pad_mode="constant", act=act, leaky=0.1, bn=bn)
    _, CH, CW, _ = conv.out_shape()
    pool = snn.pool2d_plan(ctx, N, CH, CW, OC, 3, 2, kind="max", same=True)
    fused = snn.chain_plan(ctx, [conv, pool])
    assert fused.num_steps() == 1 and "+maxpool3x3/2 in the epilogue" in fused.describe(), fused.describe()
    xt = snn.Tensor.from_numpy(ctx, x)
    got = fused(xt).numpy()
    two = pool(conv(xt)).numpy()
    want = O.pool2d(O.conv2d(x, w, b, 2, pads, "constant", act, 0.1, bn, threads=8), 3, 2, kind="max", same=True)
    assert got.shape == want.shape == two.shape
    np.testing.assert_array_equal(got, two, err_msg=fused.describe())
    np.testing.assert_allclose(got, want, err_msg=fused.describe(), rtol=1e-4, atol=1e-4)
    f, bts = fused.cost()
    assert f > 0 and bts > 0
    monkeypatch.setenv("SNNHIP_NO_STEM_POOL_FUSION", "1")
    with pytest.raises(snn.capi.SnnHipError, match="no rule matches"):  # (the chain planner reports a chain it cannot shorten; the caller keeps the two plans)
        snn.chain_plan(ctx, [conv, pool])


@pytest.mark.parametrize("kernel_fold", [True, False], ids=["kernel-fold", "fold-launches"])
@pytest.mark.parametrize("n,h,w,ic,oc,offset", [(2, 19, 45, 32, 128, 0.0), (1, 37, 70, 64, 64, 6.0), (3, 9, 33, 16, 32, 0.0), (1, 16, 64, 128, 128, 40.0), (2, 70, 300, 16, 32, 0.0),
                                                (2, 12, 40, 32, 256, 0.0), (2, 21, 37, 16, 96, 0.0)])  # (the last two: several output-channel blocks per pixel tile)
def test_wide_tile_stats_feed_instancenorm(ctx, monkeypatch, n, h, w, ic, oc, offset, kernel_fold):
    """Chain rule F on the wide kernel (the default there): reflect Pad -> Conv2D -> InstanceNorm as ONE step -- the convolution's epilogue leaves
    {mean, M2} of every output tile and channel, a fold over those records replaces the norm's statistics sweep.  Ragged tile edges in both
    directions, several images, and a layer whose mean is far from zero compared with its deviation (the tile records are merged with the
    parallel-variance update, not as raw sums).  The records are folded by the convolution kernel itself (the last block of an image to finish:
    norm_fold.h; default) or by two fold launches (SNNHIP_NO_KERNEL_FOLD).  Against the separate launches and the oracle; run twice (the
    kernel's block counters must be back at zero)."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_CONV", "wide")
    monkeypatch.setenv("SNNHIP_NORM_FUSION_MIN_MB", "0")
    if not kernel_fold:
        monkeypatch.setenv("SNNHIP_NO_KERNEL_FOLD", "1")
    x, wt = _rand((n, h, w, ic), 1), _rand((oc, ic, 3, 3), 2, 1.0 / np.sqrt(ic * 9))
    b = _rand((oc,), 3, 0.5) + offset
    beta, gamma = _rand((oc,), 4, 0.3), 1.0 + _rand((oc,), 5, 0.2)
    pad = snn.pad_plan(ctx, n, h, w, ic, (1, 1, 1, 1), "reflect")
    conv = snn.conv2d_plan(ctx, n, h + 2, w + 2, wt, b, stride=1, pads=(0, 0, 0, 0), act="", dtype=snn.F16)
    oh, ow = conv.out_shape()[1:3]  # (the reference's size rule keeps the padded extent)
    norm = snn.instancenorm_plan(ctx, n, oh, ow, oc, beta, gamma, act="relu")
    fused = snn.chain_plan(ctx, [pad, conv, norm])
    d = fused.describe()
    assert fused.num_steps() == 1 and "wide" in d and "+tile-stats" in d, d
    assert ("+tile-stats+fold" in d and "instancenorm(1 sweep)" in d) if kernel_fold else "fold of tile stats + 1 sweep" in d, d
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    y = fused(xt).numpy()
    np.testing.assert_array_equal(fused(xt).numpy(), y)
    two = norm(conv(pad(xt))).numpy()
    c = O._h(O.conv2d(O.pad(O._h(x), (1, 1, 1, 1), "reflect"), O._h(wt), b, 1, (0, 0, 0, 0), "constant", "", 0.0, None))
    want = O._h(O.instancenorm(c, beta, gamma, "relu"))
    # with an offset the fp16 STORED values are coarse (ulp of 6 is 0.004, of 40 0.03): both paths normalise the same stored tensor, the oracle a
    # differently rounded one -- the tight comparison is the one against the separate launches
    np.testing.assert_allclose(y, two, rtol=2e-3, atol=2e-3, err_msg=d)
    if offset == 0:
        np.testing.assert_allclose(y, want, err_msg=d, rtol=6e-3, atol=6e-3)
    # a DIFFERENT input right behind it through the same plan (same record / counter / mul / shift buffers): a replay of identical data cannot tell
    # fresh tile records from the previous launch's -- this one can (other scale, other mean)
    x2 = (0.5 * _rand((n, h, w, ic), 11) + 0.75).astype(np.float32)
    xt2 = snn.Tensor.from_numpy(ctx, x2, dtype=snn.F16)
    y2 = fused(xt2).numpy()
    two2 = norm(conv(pad(xt2))).numpy()
    assert np.abs(two2 - two).max() > 0.05  # (the two inputs really normalise differently)
    np.testing.assert_allclose(y2, two2, rtol=2e-3, atol=2e-3, err_msg="second input through the same plan: " + d)
    np.testing.assert_allclose(fused(xt).numpy(), y, rtol=0, atol=0, err_msg="and back to the first input")
    # the caller's convolution plan is not changed by the fusion, and the rule can be switched off
    assert "tile-stats" not in conv.describe()
    monkeypatch.setenv("SNNHIP_NORM_FUSION", "0")
    assert "tile stats" not in snn.chain_plan(ctx, [pad, conv, norm]).describe()


@pytest.mark.parametrize("order", [[3, 0], [0, 3]])
def test_wide_tile_stats_feed_instancenorm_add_at_the_end_of_a_graph_run(ctx, monkeypatch, order):
    """Rules F + H: a residual block's tail Pad -> Conv2D -> InstanceNorm -> Add(., skip).  The graph walk folds the Add into the norm (H) and offers
    the run in front + that two-input plan to the chain planner, where the convolution hands its tile statistics to the norm (F): one
    two-input chain at the Add node, conv launch + fold + ONE sweep.  The skip is 2 pixels smaller (the reference's size rule)."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_CONV", "wide")
    monkeypatch.setenv("SNNHIP_NORM_FUSION_MIN_MB", "0")
    n, h, w, c = 2, 21, 37, 64
    x, wt, b = _rand((n, h, w, c), 1), _rand((c, c, 3, 3), 2, 1.0 / np.sqrt(c * 9)), _rand((c,), 3, 0.5)
    beta, gamma = _rand((c,), 4, 0.3), 1.0 + _rand((c,), 5, 0.2)
    pre = snn.activation_plan(ctx, n, h, w, c, "relu")  # node 0: the block input (read by the run and by the Add)
    pad = snn.pad_plan(ctx, n, h, w, c, (1, 1, 1, 1), "reflect")
    conv = snn.conv2d_plan(ctx, n, h + 2, w + 2, wt, b, stride=1, pads=(0, 0, 0, 0), act="", dtype=snn.F16)
    oh, ow = conv.out_shape()[1:3]
    assert (oh, ow) == (h + 2, w + 2)
    norm = snn.instancenorm_plan(ctx, n, oh, ow, c, beta, gamma, act="")
    add = snn.add_plan(ctx, n, oh, ow, c, act="")
    nodes = [(pre, [-1], False), (pad, [0], False), (conv, [1], False), (norm, [2], False), (add, order, True)]
    fused = snn.graph_fuse(ctx, nodes)
    assert [p is None for p, _ in fused] == [False, True, True, True, False], [p.describe() if p else None for p, _ in fused]
    tail, ins = fused[4]
    d = tail.describe()
    assert ins == [0, 0] and "chain{" in d and "+tile-stats+fold" in d and "(statistics from the convolution, 1 sweep) +add" in d, (ins, d)
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    t0 = pre(xt)
    y = tail([t0, t0]).numpy()
    np.testing.assert_array_equal(tail([t0, t0]).numpy(), y)
    nrm = norm(conv(pad(t0)))
    two = add([nrm, t0] if order[0] == 3 else [t0, nrm]).numpy()
    np.testing.assert_allclose(y, two, rtol=2e-3, atol=2e-3, err_msg=d)
    t = O._h(np.maximum(O._h(x), 0))
    cc = O._h(O.conv2d(O.pad(t, (1, 1, 1, 1), "reflect"), O._h(wt), b, 1, (0, 0, 0, 0), "constant", "", 0.0, None))
    nrm = O._h(O.instancenorm(cc, beta, gamma, ""))
    want = nrm.copy() if order[0] == 3 else np.zeros_like(nrm)  # the ragged-Add rule: outside the FIRST input's extent nothing is summed
    want[:, :h, :w, :] = O._h(nrm[:, :h, :w, :] + t)
    np.testing.assert_allclose(y, want, rtol=6e-3, atol=6e-3, err_msg=d)


def test_wide_tile_stats_feed_a_normalising_convolution(ctx, monkeypatch):
    """Rules F + I: Conv2D (wide) -> InstanceNorm -> reflect Pad -> Conv2D 9x9 32 -> 3 (the tail of the style networks).  The second convolution
    normalises while it stages (I) and the norm's statistics come from the first one's tile records (F), folded by that kernel: the norm is no launch at all."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_NORM_FUSION_MIN_MB", "0")
    n, h, w, ic, c = 2, 150, 340, 64, 32  # 220 blocks of 512 pixels: large enough for the wide kernel to be the default choice
    x, w1, b1 = _rand((n, h, w, ic), 1), _rand((c, ic, 3, 3), 2, 1.0 / np.sqrt(ic * 9)), _rand((c,), 3, 0.5)
    w2, b2 = _rand((3, c, 9, 9), 6, 1.0 / np.sqrt(c * 81)), _rand((3,), 7, 0.5)
    beta, gamma = _rand((c,), 4, 0.3), 1.0 + _rand((c,), 5, 0.2)
    conv2 = snn.conv2d_plan(ctx, n, h + 8, w + 8, w2, b2, stride=1, pads=(0, 0, 0, 0), act="tanh", dtype=snn.F16)
    conv1 = snn.conv2d_plan(ctx, n, h, w, w1, b1, stride=1, pads=(1, 1, 1, 1), act="", dtype=snn.F16)
    norm = snn.instancenorm_plan(ctx, n, h, w, c, beta, gamma, act="relu")
    pad = snn.pad_plan(ctx, n, h, w, c, (4, 4, 4, 4), "reflect")
    plans = [conv1, norm, pad, conv2]
    fused = snn.chain_plan(ctx, plans)
    d = fused.describe()
    assert fused.num_steps() == 2 and "+tile-stats+fold" in d and "instancenorm(statistics from the convolution in front) -> " in d and "rowfold" in d, d
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    y = fused(xt).numpy()
    t = xt
    for pl in plans:
        t = pl(t)
    np.testing.assert_allclose(y, t.numpy(), rtol=3e-3, atol=3e-3, err_msg=d)
    cc = O._h(O.conv2d(O._h(x), O._h(w1), b1, 1, (1, 1, 1, 1), "constant", "", 0.0, None))
    r = O.pad(O._h(O.instancenorm(cc, beta, gamma, "relu")), (4, 4, 4, 4), "reflect")
    want = O._h(O.conv2d(r, O._h(w2), b2, 1, (0, 0, 0, 0), "constant", "tanh", 0.0, None))
    np.testing.assert_allclose(y, want, rtol=6e-3, atol=6e-3, err_msg=d)


@pytest.mark.parametrize("with_pad", [False, True])
@pytest.mark.parametrize("n,h,w,ic,oc,act", [(2, 19, 45, 32, 128, "relu"), (1, 37, 70, 64, 64, "leakyRelu"), (3, 9, 33, 16, 64, "relu"), (1, 16, 40, 128, 128, "")])
def test_wide_normalises_in_lds_behind_the_dma(ctx, monkeypatch, n, h, w, ic, oc, act, with_pad, with_up=False):
    """Graph rule I on the wide kernel: InstanceNorm -> [UpSampling] -> [Pad] -> Conv2D 3x3 as the norm's statistics + ONE convolution launch.  The
    DMA staging cannot transform what it copies: every thread normalises the LDS slots its own lanes wrote, before the barrier that publishes the
    chunk.  Same arithmetic and rounding point as the norm's normalise sweep: bit-identical to the separate launches; zero padding of a 'same'
    convolution stays zero; 128- and 64-channel blocks, 16- and 32-channel chunks (the planner keeps the separate launches behind a fused
    UpSampling and on 32-channel blocks, where the pass costs more than the sweep it replaces)."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_CONV", "wide")
    x = 1.5 * _rand((n, h, w, ic), 1) + 0.2
    wt, b = _rand((oc, ic, 3, 3), 2, 1.0 / np.sqrt(ic * 9)), _rand((oc,), 3, 0.5)
    beta, gamma = _rand((ic,), 4, 0.3), 1.0 + _rand((ic,), 5, 0.2)
    norm = snn.instancenorm_plan(ctx, n, h, w, ic, beta, gamma, act=act, leaky=0.1)
    plans, hh, ww = [norm], h, w
    if with_up:
        plans.append(snn.upsample_plan(ctx, n, h, w, ic, 2.0, "nearest"))
        hh, ww = 2 * h, 2 * w
    cp = (1, 1, 1, 1)
    if with_pad:
        plans.append(snn.pad_plan(ctx, n, hh, ww, ic, (1, 1, 1, 1), "reflect"))
        hh, ww, cp = hh + 2, ww + 2, (0, 0, 0, 0)
    plans.append(snn.conv2d_plan(ctx, n, hh, ww, wt, b, stride=1, pads=cp, act="relu", dtype=snn.F16))
    fused = snn.chain_plan(ctx, plans)
    d = fused.describe()
    assert fused.num_steps() == 1 and "in LDS behind the DMA) -> conv2d_mfma_wide_f16" in d, d
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    y = fused(xt).numpy()
    t = xt
    for pl in plans:
        t = pl(t)
    np.testing.assert_array_equal(y, t.numpy(), err_msg=d)
    ref = O._h(O.instancenorm(O._h(x), beta, gamma, act, 0.1))
    if with_up:
        ref = O.upsample(ref, 2.0, "nearest")
    if with_pad:
        ref = O.pad(ref, (1, 1, 1, 1), "reflect")
    want = O._h(O.conv2d(ref, O._h(wt), b, 1, cp, "constant", "relu", 0.0, None))
    np.testing.assert_allclose(y, want, err_msg=d, rtol=6e-3, atol=6e-3)
    monkeypatch.setenv("SNNHIP_NO_WIDE_NORM", "1")
    assert "in LDS behind the DMA" not in snn.chain_plan(ctx, plans).describe() if (with_pad or with_up) else True


def test_wide_residual_block_is_two_convolutions_and_one_sweep(ctx, monkeypatch):
    """Rules D + I + F + H together on a residual block of the style networks, X -> Pad -> Conv -> InstanceNorm(relu) -> Pad -> Conv -> InstanceNorm
    -> Add(., X): conv 1 leaves tile statistics (F), conv 2 normalises conv 1's output in its staging (I) and leaves tile statistics of its own
    (F), the Add is folded into the second norm's single sweep (H) -- three launches, and of the block's seven tensor passes between the
    convolutions only the last norm's read + residual read + write remain."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_CONV", "wide")
    monkeypatch.setenv("SNNHIP_NORM_FUSION_MIN_MB", "0")
    n, h, w, c = 2, 21, 37, 64
    x = _rand((n, h, w, c), 1)
    w1, b1, w2, b2 = _rand((c, c, 3, 3), 2, 1.0 / np.sqrt(c * 9)), _rand((c,), 3, 0.5), _rand((c, c, 3, 3), 6, 1.0 / np.sqrt(c * 9)), _rand((c,), 7, 0.5)
    be1, ga1, be2, ga2 = _rand((c,), 4, 0.3), 1.0 + _rand((c,), 5, 0.2), _rand((c,), 8, 0.3), 1.0 + _rand((c,), 9, 0.2)
    pre = snn.activation_plan(ctx, n, h, w, c, "relu")
    pad1 = snn.pad_plan(ctx, n, h, w, c, (1, 1, 1, 1), "reflect")
    conv1 = snn.conv2d_plan(ctx, n, h + 2, w + 2, w1, b1, stride=1, pads=(0, 0, 0, 0), act="", dtype=snn.F16)
    h1, w1_ = conv1.out_shape()[1:3]
    norm1 = snn.instancenorm_plan(ctx, n, h1, w1_, c, be1, ga1, act="relu")
    pad2 = snn.pad_plan(ctx, n, h1, w1_, c, (1, 1, 1, 1), "reflect")
    conv2 = snn.conv2d_plan(ctx, n, h1 + 2, w1_ + 2, w2, b2, stride=1, pads=(0, 0, 0, 0), act="", dtype=snn.F16)
    h2, w2_ = conv2.out_shape()[1:3]
    norm2 = snn.instancenorm_plan(ctx, n, h2, w2_, c, be2, ga2, act="")
    add = snn.add_plan(ctx, n, h2, w2_, c, act="")
    nodes = [(pre, [-1], False), (pad1, [0], False), (conv1, [1], False), (norm1, [2], False), (pad2, [3], False), (conv2, [4], False), (norm2, [5], False),
             (add, [6, 0], True)]
    fused = snn.graph_fuse(ctx, nodes)
    assert [p is None for p, _ in fused] == [False] + [True] * 6 + [False], [p.describe() if p else None for p, _ in fused]
    tail, ins = fused[7]
    d = tail.describe()
    assert ins == [0, 0] and tail.num_steps() == 3, (ins, d)
    assert d.count("+tile-stats+fold") == 2 and "instancenorm(statistics from the convolution in front) -> instancenorm(act=1, in LDS behind the DMA)" in d, d
    assert "(statistics from the convolution, 1 sweep) +add" in d, d
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    t0 = pre(xt)
    y = tail([t0, t0]).numpy()
    np.testing.assert_array_equal(tail([t0, t0]).numpy(), y)
    two = add([norm2(conv2(pad2(norm1(conv1(pad1(t0)))))), t0]).numpy()
    np.testing.assert_allclose(y, two, rtol=4e-3, atol=4e-3, err_msg=d)


def _style_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        c = int(rng.choice([32, 64, 96, 128]))  # (both layers on the wide kernel: IC % 16 == 0, OC % 32 == 0)
        c2 = int(rng.choice([32, 64, 96, 128]))
        out.append((int(rng.choice([1, 2, 3])), int(rng.integers(6, 40)), int(rng.integers(6, 75)), c, c2, str(rng.choice(["relu", "", "leakyRelu", "relu6"])),
                    str(rng.choice(["reflect", "replicate", "constant"])), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.integers(0, 1 << 30))))
    return out


@pytest.mark.parametrize("case", _style_cases(16, 20260928), ids=lambda c: "n%d_%dx%d_%d-%d_%s_%s_add%d_fold%d" % c[:9])
def test_wide_random_style_runs_match_the_separate_launches(ctx, monkeypatch, case):
    """Randomised (fixed seed) runs of the style networks' building block on the wide kernel, X -> Pad -> Conv 3x3 -> InstanceNorm(act) -> Pad -> Conv
    3x3 -> InstanceNorm [-> Add(., X)], offered to the graph walk: whatever mix of rules D / F / H / I the planner applies to the shape (tile
    statistics with the kernel's own fold or with fold launches, the norm fixed up in LDS, the Add in the norm's sweep), the result is the separate
    launches' -- bit-identical where only D / H / I apply, within fp16 rounding of the statistics where rule F supplies them."""
    import shadernn_amd as snn

    n, h, w, c, c2, act, pad_mode, with_add, kernel_fold, seed = case
    monkeypatch.setenv("SNNHIP_CONV", "wide")
    monkeypatch.setenv("SNNHIP_NORM_FUSION_MIN_MB", "0")
    if not kernel_fold:
        monkeypatch.setenv("SNNHIP_NO_KERNEL_FOLD", "1")
    x = _rand((n, h, w, c), seed)
    w1, b1 = _rand((c2, c, 3, 3), seed + 1, 1.0 / np.sqrt(c * 9)), _rand((c2,), seed + 2, 0.5)
    w2, b2 = _rand((c, c2, 3, 3), seed + 3, 1.0 / np.sqrt(c2 * 9)), _rand((c,), seed + 4, 0.5)
    be1, ga1, be2, ga2 = _rand((c2,), seed + 5, 0.3), 1.0 + _rand((c2,), seed + 6, 0.2), _rand((c,), seed + 7, 0.3), 1.0 + _rand((c,), seed + 8, 0.2)
    pre = snn.activation_plan(ctx, n, h, w, c, "tanh")
    pad1 = snn.pad_plan(ctx, n, h, w, c, (1, 1, 1, 1), pad_mode)
    conv1 = snn.conv2d_plan(ctx, n, h + 2, w + 2, w1, b1, stride=1, pads=(0, 0, 0, 0), act="", dtype=snn.F16)
    h1, w1_ = conv1.out_shape()[1:3]
    norm1 = snn.instancenorm_plan(ctx, n, h1, w1_, c2, be1, ga1, act=act, leaky=0.1)
    pad2 = snn.pad_plan(ctx, n, h1, w1_, c2, (1, 1, 1, 1), pad_mode)
    conv2 = snn.conv2d_plan(ctx, n, h1 + 2, w1_ + 2, w2, b2, stride=1, pads=(0, 0, 0, 0), act="", dtype=snn.F16)
    h2, w2_ = conv2.out_shape()[1:3]
    norm2 = snn.instancenorm_plan(ctx, n, h2, w2_, c, be2, ga2, act="")
    nodes = [(pre, [-1], False), (pad1, [0], False), (conv1, [1], False), (norm1, [2], False), (pad2, [3], False), (conv2, [4], False), (norm2, [5], not with_add)]
    if with_add:
        add = snn.add_plan(ctx, n, h2, w2_, c, act="")
        nodes.append((add, [6, 0], True))
    fused = snn.graph_fuse(ctx, nodes)
    last = len(nodes) - 1
    tail, ins = fused[last]
    d = tail.describe()
    assert all(p is None for p, _ in fused[1:last]) and "conv2d_mfma_wide_f16" in d and "tile-stats" in d, d
    assert (fused[0][0] is None) == (not with_add)  # without the Add nobody else reads the block input: its producer joins the run
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    t0 = pre(xt)
    y = (tail([t0, t0]) if with_add else tail(xt)).numpy()
    ref = norm2(conv2(pad2(norm1(conv1(pad1(t0))))))
    if with_add:
        ref = add([ref, t0])
    np.testing.assert_allclose(y, ref.numpy(), rtol=6e-3, atol=6e-3, err_msg=d)
    assert np.isfinite(y).all()


UPCONV = [(2, 13, 21, 64, 32, "relu", ""), (1, 30, 45, 128, 64, "", "2"), (1, 9, 70, 64, 64, "leakyRelu", ""), (2, 17, 33, 128, 128, "relu", "3"), (1, 2, 2, 64, 32, "", ""),
          (1, 25, 32, 64, 32, "tanh", "")]


@pytest.mark.parametrize("case", UPCONV, ids=lambda c: "x".join(map(str, c[:5])) + "_segs" + (c[6] or "auto"))
def test_upconv_low_resolution_phases_match_the_oracle_and_the_nine_tap_kernel(ctx, monkeypatch, case):
    """conv2d_upconv.hip: UpSampling2D x2 -> reflect Pad(1) -> Conv2D 3x3 as 4 phases x 2x2 pre-summed taps on the LOW-RESOLUTION tensor (forced: the
    default takes it on large maps only).  Whole output incl. the last two rows / columns, where the reference's size rule (the padded extent is the
    output extent, zeros beyond it) drops taps -- the extra low-resolution row / column with their own weight classes; ragged strips, odd extents,
    several segments and channel tiles.  Against the quantised oracle on the upsampled + padded tensor and against the 9-tap kernel on the same inputs
    (the pre-summed weights are rounded once more: compared at the fp16 tolerance, not bit for bit)."""
    import shadernn_amd as snn

    n, h, w_, ic, oc, act, segs = case
    x, wt, b = _rand((n, h, w_, ic), 1), _rand((oc, ic, 3, 3), 2, 1.0 / np.sqrt(ic * 9)), _rand((oc,), 3, 0.2)
    bn = _bn(oc, 9) if act == "tanh" else None

    def chain():
        plans = [snn.upsample_plan(ctx, n, h, w_, ic, 2.0, "nearest"), snn.pad_plan(ctx, n, 2 * h, 2 * w_, ic, (1, 1, 1, 1), "reflect"),
                 snn.conv2d_plan(ctx, n, 2 * h + 2, 2 * w_ + 2, wt, b, stride=1, pads=(0, 0, 0, 0), act=act, leaky=0.1, bn=bn, dtype=snn.F16)]
        return snn.chain_plan(ctx, plans)

    monkeypatch.setenv("SNNHIP_CONV", "upconv")
    if segs:
        monkeypatch.setenv("SNNHIP_UPCONV_SEGS", segs)
    up = chain()
    d = up.describe()
    assert up.num_steps() == 1 and "upconv" in d and "+upsample(x2)" in d and (not segs or "segments=%s x" % segs in d), d
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    y = up(xt).numpy()
    t = O.pad(O.upsample(O._h(x), 2.0, "nearest"), (1, 1, 1, 1), "reflect")
    want = O._h(O.conv2d(t, O._h(wt), b, 1, (0, 0, 0, 0), "constant", act, 0.1, bn))
    assert y.shape == want.shape == (n, 2 * h + 2, 2 * w_ + 2, oc)  # the padded extent (Q20)
    np.testing.assert_allclose(y, want, err_msg=d, **TOLH)
    np.testing.assert_allclose(y[:, -2:], want[:, -2:], err_msg="last two rows " + d, **TOLH)
    np.testing.assert_allclose(y[:, :, -2:], want[:, :, -2:], err_msg="last two columns " + d, **TOLH)
    monkeypatch.delenv("SNNHIP_CONV")
    monkeypatch.setenv("SNNHIP_CONV_UPCONV", "0")
    nine = chain()
    assert "upconv" not in nine.describe(), nine.describe()
    np.testing.assert_allclose(y, nine(xt).numpy(), err_msg=d + " vs " + nine.describe(), rtol=3e-3, atol=3e-3)


def test_upconv_presummed_weights_stay_within_their_rounding_bound_when_taps_cancel(ctx, monkeypatch):
    """conv2d_upconv.hip sums the (up to four) half-rounded taps that fall on one low-resolution pixel in fp32 and rounds the SUM to half once more;
    the 9-tap path (and the reference) multiply each half-rounded tap on its own.  The extra error per pre-summed weight is <= 2^-11 of the summed
    magnitude, so per output |upconv - nine| <= 2^-11 * sum |w_presummed| * |x| (+ the output rounding of either path).  Worst case: neighbouring
    taps that cancel, leaving a remainder much smaller than the taps -- checked here against that bound, not against a loose tolerance."""
    import shadernn_amd as snn

    n, h, w_, ic, oc = 1, 21, 27, 64, 32
    rng = np.random.default_rng(77)
    wt = rng.standard_normal((oc, ic, 3, 3)).astype(np.float32) / np.sqrt(ic * 9)
    wt[:, :, :, 1] = -wt[:, :, :, 0] + 1e-3 * rng.standard_normal((oc, ic, 3)).astype(np.float32)   # columns 0 and 1 cancel to ~1e-3 of a tap
    wt[:, :, 1, :] = -wt[:, :, 2, :] + 1e-3 * rng.standard_normal((oc, ic, 3)).astype(np.float32)   # and rows 1 and 2
    x, b = _rand((n, h, w_, ic), 5), _rand((oc,), 6, 0.2)

    def chain():
        plans = [snn.upsample_plan(ctx, n, h, w_, ic, 2.0, "nearest"), snn.pad_plan(ctx, n, 2 * h, 2 * w_, ic, (1, 1, 1, 1), "reflect"),
                 snn.conv2d_plan(ctx, n, 2 * h + 2, 2 * w_ + 2, wt, b, stride=1, pads=(0, 0, 0, 0), act="", dtype=snn.F16)]
        return snn.chain_plan(ctx, plans)

    monkeypatch.setenv("SNNHIP_CONV", "upconv")
    up = chain()
    assert "upconv" in up.describe(), up.describe()
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    y = up(xt).numpy()
    monkeypatch.delenv("SNNHIP_CONV")
    monkeypatch.setenv("SNNHIP_CONV_UPCONV", "0")
    nine = chain()
    assert "upconv" not in nine.describe(), nine.describe()
    y9 = nine(xt).numpy()
    # bound: every pre-summed weight is off by at most half an fp16 ulp of its magnitude (<= 2^-11 |w_sum|, |w_sum| <= sum of the |taps| it holds),
    # times the inputs it multiplies; plus one fp16 output rounding on each side
    t = O.pad(O.upsample(O._h(x), 2.0, "nearest"), (1, 1, 1, 1), "reflect")
    mag = O.conv2d(np.abs(t), np.abs(O._h(wt)), None, 1, (0, 0, 0, 0), "constant", "", 0.0, None)   # sum |w| |x| per output (an upper bound of sum |w_sum| |x|)
    bound = 2.0 ** -11 * mag + 2.0 ** -10 * np.maximum(np.abs(y9), 2.0 ** -14) + 1e-6
    d = np.abs(y.astype(np.float64) - y9.astype(np.float64))
    assert (d <= bound).all(), "max excess %.3e at %s" % (float((d - bound).max()), np.unravel_index(int((d - bound).argmax()), d.shape))
    want = O._h(O.conv2d(t, O._h(wt), b, 1, (0, 0, 0, 0), "constant", "", 0.0, None))
    np.testing.assert_allclose(y, want, **TOLH)


@pytest.mark.parametrize("n,h,w_,ic,oc,offset", [(2, 30, 45, 64, 32, 0.0), (1, 25, 70, 128, 64, 3.0), (3, 9, 33, 64, 64, 0.0)])
def test_upconv_block_statistics_feed_the_instancenorm_behind_it(ctx, monkeypatch, n, h, w_, ic, oc, offset):
    """Rule F on conv2d_upconv.hip: UpSampling -> Pad -> Conv2D -> InstanceNorm as the convolution (whose blocks leave {pixels, sum, sum of squares} records
    around the channel's bias and whose last block per image folds them into the norm's shift / mul) + ONE normalise sweep.  Several images, a mean far
    from zero, several row segments per strip; against the separate launches and the oracle; a second, different input through the same plan."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_CONV", "upconv")
    monkeypatch.setenv("SNNHIP_NORM_FUSION_MIN_MB", "0")
    monkeypatch.setenv("SNNHIP_UPCONV_SEGS", "2")
    x, wt = _rand((n, h, w_, ic), 1), _rand((oc, ic, 3, 3), 2, 1.0 / np.sqrt(ic * 9))
    b = _rand((oc,), 3, 0.5) + offset
    beta, gamma = _rand((oc,), 4, 0.3), 1.0 + _rand((oc,), 5, 0.2)
    up = snn.upsample_plan(ctx, n, h, w_, ic, 2.0, "nearest")
    pad = snn.pad_plan(ctx, n, 2 * h, 2 * w_, ic, (1, 1, 1, 1), "reflect")
    conv = snn.conv2d_plan(ctx, n, 2 * h + 2, 2 * w_ + 2, wt, b, stride=1, pads=(0, 0, 0, 0), act="", dtype=snn.F16)
    norm = snn.instancenorm_plan(ctx, n, 2 * h + 2, 2 * w_ + 2, oc, beta, gamma, act="relu")
    fused = snn.chain_plan(ctx, [up, pad, conv, norm])
    d = fused.describe()
    assert fused.num_steps() == 1 and "upconv" in d and "+tile-stats+fold" in d and "instancenorm(1 sweep)" in d, d
    for seed in (1, 11):
        xs = x if seed == 1 else (0.5 * _rand((n, h, w_, ic), seed) + 0.75).astype(np.float32)
        xt = snn.Tensor.from_numpy(ctx, xs, dtype=snn.F16)
        y = fused(xt).numpy()
        np.testing.assert_array_equal(fused(xt).numpy(), y)
        two = norm(snn.chain_plan(ctx, [up, pad, conv])(xt)).numpy()
        np.testing.assert_allclose(y, two, rtol=2e-3, atol=2e-3, err_msg="seed %d: %s" % (seed, d))
        if offset == 0:
            t = O.pad(O.upsample(O._h(xs), 2.0, "nearest"), (1, 1, 1, 1), "reflect")
            c = O._h(O.conv2d(t, O._h(wt), b, 1, (0, 0, 0, 0), "constant", "", 0.0, None))
            np.testing.assert_allclose(y, O._h(O.instancenorm(c, beta, gamma, "relu")), err_msg=d, rtol=6e-3, atol=6e-3)


@pytest.mark.parametrize("n,h,w_,oc,act,segs", [(2, 30, 45, 32, "relu", "2"), (1, 26, 70, 64, "", "1"), (3, 24, 33, 32, "leakyRelu", "3")])
def test_upconv_normalises_the_low_resolution_tensor_while_it_stages(ctx, monkeypatch, n, h, w_, oc, act, segs):
    """Graph rule I on conv2d_upconv.hip (round 6): InstanceNorm [-> ReLU] -> UpSampling x2 -> reflect Pad(1) -> Conv2D 3x3 (64 channels in) as the norm's
    statistics sweep + fold and ONE up-convolution launch that applies half(act(x * mul + shift)) to the LOW-RESOLUTION pixels it stages.  Same arithmetic
    and rounding point as the norm's own normalise sweep: bit-identical to the separate launches; within tolerance of the oracle; several images (the
    shift / mul of the block's own image), several row segments, border strips; with the norm of rule F behind it as well (Candy's 64 -> 32 layer)."""
    import shadernn_amd as snn

    ic = 64
    monkeypatch.setenv("SNNHIP_CONV", "upconv")
    monkeypatch.setenv("SNNHIP_UPCONV_SEGS", segs)
    monkeypatch.setenv("SNNHIP_NORM_FUSION_MIN_MB", "0")
    x = 1.5 * _rand((n, h, w_, ic), 1) + 0.2
    x[1:] *= 0.5  # (other statistics per image)
    wt, b = _rand((oc, ic, 3, 3), 2, 1.0 / np.sqrt(ic * 9)), _rand((oc,), 3, 0.5)
    beta, gamma = _rand((ic,), 4, 0.3), 1.0 + _rand((ic,), 5, 0.2)
    norm = snn.instancenorm_plan(ctx, n, h, w_, ic, beta, gamma, act=act, leaky=0.1)
    up = snn.upsample_plan(ctx, n, h, w_, ic, 2.0, "nearest")
    pad = snn.pad_plan(ctx, n, 2 * h, 2 * w_, ic, (1, 1, 1, 1), "reflect")
    conv = snn.conv2d_plan(ctx, n, 2 * h + 2, 2 * w_ + 2, wt, b, stride=1, pads=(0, 0, 0, 0), act="", dtype=snn.F16)
    fused = snn.chain_plan(ctx, [norm, up, pad, conv])
    d = fused.describe()
    assert fused.num_steps() == 1 and "instancenorm(statistics sweep + fold) -> instancenorm(act=" in d and "in the staging" in d and "upconv" in d, d
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    y = fused(xt).numpy()
    np.testing.assert_array_equal(fused(xt).numpy(), y)
    apart = snn.chain_plan(ctx, [up, pad, conv])
    assert "upconv" in apart.describe() and "in the staging" not in apart.describe()
    np.testing.assert_array_equal(y, apart(norm(xt)).numpy())
    ref = O.pad(O.upsample(O._h(O.instancenorm(O._h(x), beta, gamma, act, 0.1)), 2.0, "nearest"), (1, 1, 1, 1), "reflect")
    want = O._h(O.conv2d(ref, O._h(wt), b, 1, (0, 0, 0, 0), "constant", "", 0.0, None))
    np.testing.assert_allclose(y, want, err_msg=d, rtol=6e-3, atol=6e-3)
    # ... and with rule F behind it: the same launch also leaves the statistics records of the norm that follows
    beta2, gamma2 = _rand((oc,), 6, 0.3), 1.0 + _rand((oc,), 7, 0.2)
    norm2 = snn.instancenorm_plan(ctx, n, 2 * h + 2, 2 * w_ + 2, oc, beta2, gamma2, act="relu")
    both = snn.chain_plan(ctx, [norm, up, pad, conv, norm2])
    d2 = both.describe()
    assert "in the staging" in d2 and "+tile-stats+fold" in d2, d2
    y2 = both(xt).numpy()
    np.testing.assert_allclose(y2, norm2(apart(norm(xt))).numpy(), rtol=2e-3, atol=2e-3, err_msg=d2)
    # the switch: SNNHIP_NO_UPCONV_NORM keeps the norm's own normalise sweep in front of the up-convolution
    monkeypatch.setenv("SNNHIP_NO_UPCONV_NORM", "1")
    assert "in the staging" not in snn.chain_plan(ctx, [norm, up, pad, conv]).describe()


def test_an_oversize_up_convolution_is_a_description_only_plan_that_runs_fused(ctx):
    """Candy's 64 -> 32 up-convolution at bench.py's c5 micro-batch of 32: the tensor the layer nominally reads -- the x2-upsampled, padded 64-channel map --
    has 32 x 818 x 1378 x 64 = 2.3e9 elements, beyond the 32-bit element offsets of every fp16 convolution kernel, and it never exists: graph rule D
    evaluates the layer on the low-resolution tensor (conv2d_upconv).  The per-layer plan the host creates first is therefore description-only (geometry +
    weights for the chain planner): running it fails with a message, the fused chain runs and gives every image what a two-image batch gives."""
    import shadernn_amd as snn

    n, h, w_, ic, oc = 32, 408, 688, 64, 32
    wt, b = _rand((oc, ic, 3, 3), 2, 1.0 / np.sqrt(ic * 9)), _rand((oc,), 3, 0.2)

    def plans(nn):
        return [snn.upsample_plan(ctx, nn, h, w_, ic, 2.0, "nearest"), snn.pad_plan(ctx, nn, 2 * h, 2 * w_, ic, (1, 1, 1, 1), "reflect"),
                snn.conv2d_plan(ctx, nn, 2 * h + 2, 2 * w_ + 2, wt, b, stride=1, pads=(0, 0, 0, 0), act="", dtype=snn.F16)]

    big = plans(n)
    assert "OVERSIZE" in big[2].describe(), big[2].describe()
    pair = _rand((2, h, w_, ic), 1)
    with pytest.raises(snn.SnnHipError, match="only runs fused"):
        big[2](snn.Tensor(ctx, 1, 4, 4, ic, dtype=snn.F16), snn.Tensor(ctx, 1, 4, 4, oc, dtype=snn.F16))  # (the plan refuses before it looks at a tensor)
    fused = snn.chain_plan(ctx, big)
    assert fused.num_steps() == 1 and "upconv" in fused.describe() and "+upsample(x2)" in fused.describe(), fused.describe()
    x = np.empty((n, h, w_, ic), np.float32)
    x[0::2], x[1::2] = pair[0], pair[1]
    y = fused(snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)).numpy()
    del x
    small = snn.chain_plan(ctx, plans(2))
    assert "OVERSIZE" not in small.describe()
    want = small(snn.Tensor.from_numpy(ctx, pair, dtype=snn.F16)).numpy()
    assert y.shape == (n, 2 * h + 2, 2 * w_ + 2, oc) and np.isfinite(want).all()
    for i in (0, 1, 16, 31):
        np.testing.assert_array_equal(y[i], want[i % 2], err_msg="image %d" % i)
