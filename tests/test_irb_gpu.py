"""irb_fused.hip (chain rule G): Conv2D 1x1 -> DepthwiseConv2D 3x3 -> Conv2D 1x1 [-> Add with the block input] as one kernel, against the CPU
oracle run layer by layer and against the three / four separate HIP layers: MobileNetV2's block shapes (BASELINE configs[3]), ragged extents,
channel counts that are not multiples of 16, both strides."""
import os

import numpy as np
import pytest

import oracle_lib as O
from test_ops_gpu import _bn, _rand

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-4)

# N, H, W, C, Ch, Co, stride, residual, (act1, act2, act3)
CASES = [(2, 56, 56, 24, 144, 24, 1, True, ("relu6", "relu6", "")),      # MobileNetV2 b02
         (2, 112, 112, 16, 96, 24, 2, False, ("relu6", "relu6", "")),    # b01 (stride 2, 8x8 tiles with 17x17 halo)
         (3, 28, 28, 32, 192, 64, 2, False, ("relu6", "relu6", "")),     # b06
         (2, 14, 14, 64, 384, 64, 1, True, ("relu6", "relu6", "")),      # b07 (ragged 14 = 8 + 6 rows)
         (2, 14, 14, 64, 448, 160, 2, False, ("relu6", "relu6", "")),    # b13-like (the real b13, 96 -> 576 -> 160 s2: its 15x15x96 x tile does not fit LDS)
         (3, 7, 7, 160, 960, 160, 1, True, ("relu6", "relu6", "")),      # b14
         (2, 7, 7, 160, 960, 320, 1, False, ("relu6", "relu6", "")),     # b16 (20 output blocks)
         (1, 19, 23, 8, 40, 12, 1, False, ("relu", "leakyRelu", "relu")),  # odd extents, Ch % 16 = 8, Co % 16 = 12
         (2, 9, 33, 12, 56, 12, 1, True, ("", "relu6", "")),
         (1, 21, 17, 20, 72, 36, 2, False, ("relu6", "", "leakyRelu"))]


def _layers(case, seed):
    N, H, W, C, Ch, Co, s, res, acts = case
    we, be, bne = _rand((Ch, C, 1, 1), seed, 1.0 / np.sqrt(C)), _rand((Ch,), seed + 1, 0.1), _bn(Ch, seed + 2)
    wd, bd, bnd = _rand((Ch, 3, 3), seed + 3, 1.0 / 3.0), _rand((Ch,), seed + 4, 0.1), _bn(Ch, seed + 5)
    wp, bp, bnp = _rand((Co, Ch, 1, 1), seed + 6, 1.0 / np.sqrt(Ch)), _rand((Co,), seed + 7, 0.1), _bn(Co, seed + 8)
    return (we, be, bne), (wd, bd, bnd), (wp, bp, bnp)


def _oracle(case, x, L):
    N, H, W, C, Ch, Co, s, res, acts = case
    (we, be, bne), (wd, bd, bnd), (wp, bp, bnp) = L
    h = O.conv2d(x, we, be, 1, (0, 0, 0, 0), "constant", acts[0], 0.1, bne, threads=8)
    d = O.depthwise(h, wd, bd, s, O.padding_offsets("same", 3), acts[1], 0.1, bnd)
    y = O.conv2d(d, wp, bp, 1, (0, 0, 0, 0), "constant", acts[2], 0.1, bnp, threads=8)
    return O.add_act(y, x, "") if res else y


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%dx%d_%d-%d-%d_s%d%s" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], "_res" if c[7] else ""))
def test_irb_fused_matches_oracle_and_separate_layers(ctx, case, monkeypatch):
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_IRB_FUSION", "all")  # by default only blocks on >= 28x28 maps fuse (where it is faster); the kernel takes them all

    N, H, W, C, Ch, Co, s, res, acts = case
    x = _rand((N, H, W, C), 71)
    L = _layers(case, 80)
    (we, be, bne), (wd, bd, bnd), (wp, bp, bnp) = L
    pe = snn.conv2d_plan(ctx, N, H, W, we, be, act=acts[0], leaky=0.1, bn=bne)
    pd = snn.conv2d_plan(ctx, N, H, W, wd, bd, stride=s, pads=O.padding_offsets("same", 3), act=acts[1], leaky=0.1, bn=bnd, depthwise=True)
    _, OH, OW, _ = pd.out_shape()
    pp = snn.conv2d_plan(ctx, N, OH, OW, wp, bp, act=acts[2], leaky=0.1, bn=bnp)
    xt = snn.Tensor.from_numpy(ctx, x)
    sep = pp(pd(pe(xt)))
    want = _oracle(case, x, L)
    if res:
        pa = snn.add_plan(ctx, N, OH, OW, Co, act="")
        sep = pa([sep, xt])
        fused = snn.graph_fuse(ctx, [(pe, [-1], False), (pd, [0], False), (pp, [1], False), (pa, [2, -1], True)])
        assert [f[0] is None for f in fused] == [True, True, True, False], [f[0] and f[0].describe() for f in fused]
        plan, ins = fused[3]
        assert ins == [-1]
    else:
        plan = snn.chain_plan(ctx, [pe, pd, pp])
        assert plan.num_steps() == 1
    assert "irb_fused" in plan.describe() and ("+ add" in plan.describe()) == res, plan.describe()
    got = plan(xt).numpy()
    assert got.shape == want.shape == (N, OH, OW, Co)
    np.testing.assert_allclose(got, want, err_msg=plan.describe(), **TOL)
    np.testing.assert_allclose(got, sep.numpy(), err_msg=plan.describe(), rtol=2e-5, atol=2e-5)
    f, b = plan.cost()
    assert f > 0 and b > 0


def test_irb_fusion_can_be_switched_off_and_declines_what_it_cannot_do(ctx, monkeypatch):
    import shadernn_amd as snn

    case = (1, 32, 32, 16, 64, 16, 1, False, ("relu6", "relu6", ""))
    (we, be, bne), (wd, bd, bnd), (wp, bp, bnp) = _layers(case, 90)
    pe = snn.conv2d_plan(ctx, 1, 32, 32, we, be, act="relu6", bn=bne)
    pd = snn.conv2d_plan(ctx, 1, 32, 32, wd, bd, pads=(1, 1, 1, 1), act="relu6", bn=bnd, depthwise=True)
    pp = snn.conv2d_plan(ctx, 1, 32, 32, wp, bp, bn=bnp)
    assert "irb_fused" in snn.chain_plan(ctx, [pe, pd, pp]).describe()
    small = [snn.conv2d_plan(ctx, 1, 12, 12, we, be, act="relu6", bn=bne), snn.conv2d_plan(ctx, 1, 12, 12, wd, bd, pads=(1, 1, 1, 1), act="relu6", bn=bnd, depthwise=True),
             snn.conv2d_plan(ctx, 1, 12, 12, wp, bp, bn=bnp)]
    with pytest.raises(snn.SnnHipError):  # small maps stay unfused by default (the separate layers are faster there) ...
        snn.chain_plan(ctx, small)
    monkeypatch.setenv("SNNHIP_IRB_FUSION", "all")  # ... unless asked for
    assert "irb_fused" in snn.chain_plan(ctx, small).describe()
    monkeypatch.delenv("SNNHIP_IRB_FUSION")
    monkeypatch.setenv("SNNHIP_NO_IRB_FUSION", "1")
    with pytest.raises(snn.SnnHipError) as e:
        snn.chain_plan(ctx, [pe, pd, pp])
    assert e.value.code == snn.E_UNSUPPORTED
    monkeypatch.delenv("SNNHIP_NO_IRB_FUSION")
    pt = snn.conv2d_plan(ctx, 1, 32, 32, we, be, act="tanh", bn=bne)  # a non-"simple" activation: the expand layer stays a launch of its own ...
    two = snn.chain_plan(ctx, [pt, pd, pp])                            # ... and the depthwise -> pointwise pair behind it still fuses
    assert two.num_steps() == 2 and "[depthwise3x3" in two.describe(), two.describe()


@pytest.mark.parametrize("mode", ["default", "all"])
def test_mobilenetv2_graph_uses_the_fused_blocks(ctx, monkeypatch, mode):
    """MobileNetV2 through GraphRunner (snnhip_graph_fuse): the inverted-residual blocks run as ONE kernel each -- by default those on maps of
    28x28 and larger (b01-b06 at 224x224), with SNNHIP_IRB_FUSION=all every block that fits LDS; the result equals the unfused graph and the oracle."""
    import shadernn_amd as snn
    from shadernn_amd import models

    if mode == "all":
        monkeypatch.setenv("SNNHIP_IRB_FUSION", "all")
    net = models.mobilenetv2(seed=3, num_classes=10)
    x = np.random.default_rng(4).random((2, 96, 96, 3), dtype=np.float32)
    r = snn.GraphRunner(ctx, net, 2, 96, 96)
    y = r(x)
    fused = [d for d in r.describe() if "irb_fused" in d]
    if mode == "all":
        assert 14 <= len(fused) <= 16 and sum("+ add" in d for d in fused) == 10, r.describe()
    else:
        assert len(fused) == 1 and "16->96" in fused[0], r.describe()  # at 96x96 only b01 (48x48) is on a large map
    want = O.forward(net, x, threads=8)
    np.testing.assert_allclose(y.reshape(2, -1), want.reshape(2, -1), **TOL)
    y0 = snn.GraphRunner(ctx, net, 2, 96, 96, fuse=False)(x)
    np.testing.assert_allclose(y, y0, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("G", [1, 2, 4])
@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[7]], ids=["b02", "b01", "ragged"])
def test_irb_every_wave_tile_size_matches_the_separate_layers(ctx, monkeypatch, case, G):
    """The kernel's three tile sizes per wave (SNNHIP_IRB_WAVE_G = 1 | 2 | 4: 2x8, 4x8, 8x8 pixels; the default is 2 for stride 1 and 1 for stride
    2) on the same inputs: tile decode, halo extents and the wave-per-block choice differ, the results must not."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_IRB_FUSION", "all")
    monkeypatch.setenv("SNNHIP_IRB_WAVE_G", str(G))
    N, H, W, C, Ch, Co, s, res, acts = case
    x = _rand((N, H, W, C), 71)
    (we, be, bne), (wd, bd, bnd), (wp, bp, bnp) = _layers(case, 80)
    pe = snn.conv2d_plan(ctx, N, H, W, we, be, act=acts[0], leaky=0.1, bn=bne)
    pd = snn.conv2d_plan(ctx, N, H, W, wd, bd, stride=s, pads=O.padding_offsets("same", 3), act=acts[1], leaky=0.1, bn=bnd, depthwise=True)
    _, OH, OW, _ = pd.out_shape()
    pp = snn.conv2d_plan(ctx, N, OH, OW, wp, bp, act=acts[2], leaky=0.1, bn=bnp)
    xt = snn.Tensor.from_numpy(ctx, x)
    sep = pp(pd(pe(xt)))
    if res:
        pa = snn.add_plan(ctx, N, OH, OW, Co, act="")
        sep = pa([sep, xt])
        plan = snn.graph_fuse(ctx, [(pe, [-1], False), (pd, [0], False), (pp, [1], False), (pa, [2, -1], True)])[3][0]
    else:
        plan = snn.chain_plan(ctx, [pe, pd, pp])
    assert "tile=%dx8px per wave" % (2 * G) in plan.describe(), plan.describe()
    np.testing.assert_allclose(plan(xt).numpy(), sep.numpy(), err_msg=plan.describe(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("case", [(2, 112, 112, 32, 16, 1, "relu6", ""), (1, 45, 37, 16, 24, 2, "relu6", "relu"), (2, 30, 30, 48, 8, 1, "", "")],
                         ids=["mbv2_block0", "s2_ragged", "c48"])
def test_depthwise_pointwise_pair_runs_as_one_kernel(ctx, case):
    """Rule G without an expand layer: DepthwiseConv2D 3x3 -> Conv2D 1x1 (MobileNetV2's first block, expansion factor 1): the kernel's hidden slice is
    the x tile itself.  Against the oracle and the two separate layers."""
    import shadernn_amd as snn

    N, H, W, C, Co, s, a1, a2 = case
    x = _rand((N, H, W, C), 3)
    wd, bd, bnd = _rand((C, 3, 3), 4, 1.0 / 3.0), _rand((C,), 5, 0.1), _bn(C, 6)
    wp, bp, bnp = _rand((Co, C, 1, 1), 7, 1.0 / np.sqrt(C)), _rand((Co,), 8, 0.1), _bn(Co, 9)
    pd = snn.conv2d_plan(ctx, N, H, W, wd, bd, stride=s, pads=O.padding_offsets("same", 3), act=a1, bn=bnd, depthwise=True)
    _, OH, OW, _ = pd.out_shape()
    pp = snn.conv2d_plan(ctx, N, OH, OW, wp, bp, act=a2, bn=bnp)
    plan = snn.chain_plan(ctx, [pd, pp])
    assert plan.num_steps() == 1 and "irb_fused" in plan.describe() and "[depthwise3x3 %d s%d + conv1x1" % (C, s) in plan.describe(), plan.describe()
    xt = snn.Tensor.from_numpy(ctx, x)
    got = plan(xt).numpy()
    d = O.depthwise(x, wd, bd, s, O.padding_offsets("same", 3), a1, 0.0, bnd)
    want = O.conv2d(d, wp, bp, 1, (0, 0, 0, 0), "constant", a2, 0.0, bnp, threads=8)
    np.testing.assert_allclose(got, want, err_msg=plan.describe(), **TOL)
    np.testing.assert_allclose(got, pp(pd(xt)).numpy(), err_msg=plan.describe(), rtol=2e-5, atol=2e-5)


# N, IH, IW, stem stride, stem pads ("same" | "none"), C (stem channels), Co, depthwise stride, (stem act, dw act, pw act), wave tile G
STEM_CASES = [(2, 224, 224, 2, "same", 32, 16, 1, ("relu6", "relu6", ""), None),     # MobileNetV2's head
              (1, 75, 61, 2, "same", 32, 16, 1, ("relu6", "relu6", ""), "4"),        # ragged, odd image, 8x8 wave tiles
              (2, 63, 70, 1, "same", 16, 24, 2, ("relu", "leakyRelu", "relu"), None),  # stride-1 stem, stride-2 depthwise
              (1, 64, 64, 2, "none", 48, 8, 1, ("", "relu6", ""), "1")]              # no padding at all in the stem, 3 slices


@pytest.mark.parametrize("case", STEM_CASES, ids=["mbv2_head", "ragged_G4", "s1_stem_s2_dw", "nopad_48"])
def test_stem_depthwise_pointwise_runs_as_one_kernel(ctx, case, monkeypatch):
    """Rule G with the network's stem as the 'expand' layer: Conv2D 3x3 (3 -> C) -> DepthwiseConv2D 3x3 -> Conv2D 1x1 as one launch (the staging gathers the
    27 image values per pixel, the stem's output never reaches memory).  Against the oracle layer by layer and the three separate layers; the switch
    rule is opt-in (SNNHIP_STEM_IRB_FUSION=1: no faster than the stem + two-layer kernel at the benchmark size)."""
    import shadernn_amd as snn

    N, IH, IW, ss, spad, C, Co, s, acts, G = case
    monkeypatch.setenv("SNNHIP_STEM_IRB_FUSION", "1")
    if G:
        monkeypatch.setenv("SNNHIP_IRB_WAVE_G", G)
    x = _rand((N, IH, IW, 3), 21)
    ws, bs, bns = _rand((C, 3, 3, 3), 22, 1.0 / np.sqrt(27)), _rand((C,), 23, 0.1), _bn(C, 24)
    wd, bd, bnd = _rand((C, 3, 3), 25, 1.0 / 3.0), _rand((C,), 26, 0.1), _bn(C, 27)
    wp, bp, bnp = _rand((Co, C, 1, 1), 28, 1.0 / np.sqrt(C)), _rand((Co,), 29, 0.1), _bn(Co, 30)
    spads = O.padding_offsets("same", 3) if spad == "same" else (0, 0, 0, 0)
    smode = "constant" if spad == "same" else "none"
    ps = snn.conv2d_plan(ctx, N, IH, IW, ws, bs, stride=ss, pads=spads, pad_mode=smode, act=acts[0], leaky=0.1, bn=bns)
    _, H, W, _ = ps.out_shape()
    pd = snn.conv2d_plan(ctx, N, H, W, wd, bd, stride=s, pads=O.padding_offsets("same", 3), act=acts[1], leaky=0.1, bn=bnd, depthwise=True)
    _, OH, OW, _ = pd.out_shape()
    pp = snn.conv2d_plan(ctx, N, OH, OW, wp, bp, act=acts[2], leaky=0.1, bn=bnp)
    plan = snn.chain_plan(ctx, [ps, pd, pp])
    assert plan.num_steps() == 1 and "irb_fused" in plan.describe() and "[stem conv3x3 s%d 3->%d + depthwise3x3 s%d + conv1x1" % (ss, C, s) in plan.describe(), plan.describe()
    xt = snn.Tensor.from_numpy(ctx, x)
    got = plan(xt).numpy()
    h = O.conv2d(x, ws, bs, ss, spads, smode, acts[0], 0.1, bns, threads=8)
    d = O.depthwise(h, wd, bd, s, O.padding_offsets("same", 3), acts[1], 0.1, bnd)
    want = O.conv2d(d, wp, bp, 1, (0, 0, 0, 0), "constant", acts[2], 0.1, bnp, threads=8)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, err_msg=plan.describe(), **TOL)
    np.testing.assert_allclose(got, pp(pd(ps(xt))).numpy(), err_msg=plan.describe(), rtol=2e-5, atol=2e-5)
    f, b = plan.cost()
    assert f > 0 and b > 0
    monkeypatch.delenv("SNNHIP_STEM_IRB_FUSION")
    two = snn.chain_plan(ctx, [ps, pd, pp])
    assert two.num_steps() == 2 and "stem conv3x3" not in two.describe(), two.describe()
    np.testing.assert_allclose(two(xt).numpy(), got, rtol=2e-5, atol=2e-5)


# N, H, W, C, Co, depthwise activation, pointwise activation
MARCH_CASES = [(2, 112, 112, 32, 16, "relu6", ""),       # MobileNetV2's first block (one 16-pixel group per compute wave)
               (1, 112, 112, 32, 16, "relu6", ""),       # one image: 14 row runs of 8 rows, the runs' halo rows come from the neighbouring runs' rows
               (1, 45, 37, 16, 24, "relu6", "relu"),     # ragged width (3 groups, the last one 5 pixels), two output blocks with 8 live channels in the second
               (3, 30, 70, 32, 8, "", "leakyRelu"),      # half an output block
               (1, 9, 128, 32, 32, "relu", ""),          # 8 groups on 7 compute waves, 20 DMA pieces per row
               (2, 2, 19, 16, 4, "relu6", ""),           # fewer rows than the loader keeps in flight
               (1, 1, 16, 32, 16, "", "")]               # a single row


@pytest.mark.parametrize("case", MARCH_CASES, ids=lambda c: "%dx%dx%dx%d-%d" % c[:5])
def test_depthwise_pointwise_row_marching_kernel(ctx, case, monkeypatch):
    """dwpw_march.hip (forced; the default takes it on large maps only): DepthwiseConv2D 3x3 stride 1 -> Conv2D 1x1 as a loader wave + seven compute
    waves marching down the rows.  Whole output against the oracle, against the two separate layers, and against the tile kernel it replaces
    (SNNHIP_DWPW_MARCH=0: irb_fused's no-expand mode, same inputs)."""
    import shadernn_amd as snn

    N, H, W, C, Co, a1, a2 = case
    x = _rand((N, H, W, C), 13)
    wd, bd, bnd = _rand((C, 3, 3), 14, 1.0 / 3.0), _rand((C,), 15, 0.1), _bn(C, 16)
    wp, bp, bnp = _rand((Co, C, 1, 1), 17, 1.0 / np.sqrt(C)), _rand((Co,), 18, 0.1), (_bn(Co, 19) if a2 != "leakyRelu" else None)
    pd = snn.conv2d_plan(ctx, N, H, W, wd, bd, stride=1, pads=O.padding_offsets("same", 3), act=a1, leaky=0.1, bn=bnd, depthwise=True)
    pp = snn.conv2d_plan(ctx, N, H, W, wp, bp, act=a2, leaky=0.1, bn=bnp)
    monkeypatch.setenv("SNNHIP_DWPW_MARCH", "1")
    plan = snn.chain_plan(ctx, [pd, pp])
    d = plan.describe()
    assert plan.num_steps() == 1 and "dwpw_march_f32 [depthwise3x3 %d s1 + conv1x1 %d->%d]" % (C, C, Co) in d, d
    xt = snn.Tensor.from_numpy(ctx, x)
    got = plan(xt).numpy()
    h = O.depthwise(x, wd, bd, 1, O.padding_offsets("same", 3), a1, 0.1, bnd)
    want = O.conv2d(h, wp, bp, 1, (0, 0, 0, 0), "constant", a2, 0.1, bnp, threads=8)
    assert got.shape == want.shape == (N, H, W, Co)
    np.testing.assert_allclose(got, want, err_msg=d, **TOL)
    np.testing.assert_allclose(got, pp(pd(xt)).numpy(), err_msg=d, rtol=2e-5, atol=2e-5)
    np.testing.assert_array_equal(plan(xt).numpy(), got)   # a second run through the same ring: bit-identical
    monkeypatch.setenv("SNNHIP_DWPW_MARCH", "0")
    if H * W >= 28 * 28:  # (the tile kernel declines smaller maps: there the pair stays two launches)
        tile = snn.chain_plan(ctx, [pd, pp])
        assert "irb_fused" in tile.describe(), tile.describe()
        np.testing.assert_allclose(got, tile(xt).numpy(), err_msg=d + " vs " + tile.describe(), rtol=2e-5, atol=2e-5)


def test_depthwise_pointwise_default_rule_takes_the_marching_kernel_on_large_maps_only(ctx, monkeypatch):
    import shadernn_amd as snn

    monkeypatch.delenv("SNNHIP_DWPW_MARCH", raising=False)

    def kind(N, H, W, C, Co, s=1):
        wd, wp = _rand((C, 3, 3), 1), _rand((Co, C, 1, 1), 2)
        pd = snn.conv2d_plan(ctx, N, H, W, wd, None, stride=s, pads=O.padding_offsets("same", 3), act="relu6", depthwise=True)
        _, OH, OW, _ = pd.out_shape()
        return snn.chain_plan(ctx, [pd, snn.conv2d_plan(ctx, N, OH, OW, wp, None)]).describe()

    assert "dwpw_march" in kind(64, 112, 112, 32, 16)       # 103 MB of input: a stream
    assert "irb_fused" in kind(2, 112, 112, 32, 16)          # 3 MB: the tile kernel
    assert "irb_fused" in kind(64, 112, 112, 32, 16, s=2)    # stride 2 is not the marching kernel's
    assert "irb_fused" in kind(64, 112, 112, 48, 16)         # nor are 48 channels
    assert "irb_fused" in kind(32, 112, 112, 64, 16)         # ... or 64 (the operands of a lane would not fit its registers)


# N, IH, IW, Co, (stem act, dw act, pw act)
STEM_MARCH_CASES = [(2, 224, 224, 16, ("relu6", "relu6", "")),      # MobileNetV2's head (3 DMA pieces per image row, one 16-pixel group per compute wave)
                    (1, 224, 224, 16, ("relu6", "relu6", "")),      # one image: 14 row runs, every run recomputes its two halo stem rows
                    (1, 64, 96, 8, ("relu", "leakyRelu", "relu")),  # 2 pieces per image row, 3 groups, half an output block
                    (3, 36, 40, 12, ("", "relu6", "")),             # ragged: 20 output columns (2 groups, the second one 4 pixels), 18 rows
                    (1, 4, 256, 16, ("relu6", "", "")),             # 128 columns = 8 groups on 7 waves, two output rows
                    (2, 2, 16, 4, ("relu6", "relu6", ""))]          # a single output row


@pytest.mark.parametrize("case", STEM_MARCH_CASES, ids=lambda c: "%dx%dx%d-%d" % c[:4])
def test_stem_depthwise_pointwise_row_marching_kernel(ctx, case, monkeypatch):
    """stem_dwpw_march_kernel (forced; default on large maps): Conv2D 3x3 stride 2 (RGB -> 32) -> DepthwiseConv2D 3x3 -> Conv2D 1x1 as one launch, the stem's
    output only ever in the LDS ring of stem rows.  Whole output against the oracle layer by layer, the three separate layers, and the two-launch form."""
    import shadernn_amd as snn

    N, IH, IW, Co, acts = case
    C = 32
    x = _rand((N, IH, IW, 3), 41)
    ws, bs, bns = _rand((C, 3, 3, 3), 42, 1.0 / np.sqrt(27)), _rand((C,), 43, 0.1), _bn(C, 44)
    wd, bd, bnd = _rand((C, 3, 3), 45, 1.0 / 3.0), _rand((C,), 46, 0.1), _bn(C, 47)
    wp, bp, bnp = _rand((Co, C, 1, 1), 48, 1.0 / np.sqrt(C)), _rand((Co,), 49, 0.1), _bn(Co, 50)
    same = O.padding_offsets("same", 3)
    ps = snn.conv2d_plan(ctx, N, IH, IW, ws, bs, stride=2, pads=same, act=acts[0], leaky=0.1, bn=bns)
    _, H, W, _ = ps.out_shape()
    assert (H, W) == (IH // 2, IW // 2)
    pd = snn.conv2d_plan(ctx, N, H, W, wd, bd, stride=1, pads=same, act=acts[1], leaky=0.1, bn=bnd, depthwise=True)
    pp = snn.conv2d_plan(ctx, N, H, W, wp, bp, act=acts[2], leaky=0.1, bn=bnp)
    monkeypatch.setenv("SNNHIP_DWPW_MARCH", "1")
    plan = snn.chain_plan(ctx, [ps, pd, pp])
    d = plan.describe()
    assert plan.num_steps() == 1 and "stem_dwpw_march_f32 [stem conv3x3 s2 3->32 + depthwise3x3 32 s1 + conv1x1 32->%d]" % Co in d, d
    xt = snn.Tensor.from_numpy(ctx, x)
    got = plan(xt).numpy()
    h = O.conv2d(x, ws, bs, 2, same, "constant", acts[0], 0.1, bns, threads=8)
    dd = O.depthwise(h, wd, bd, 1, same, acts[1], 0.1, bnd)
    want = O.conv2d(dd, wp, bp, 1, (0, 0, 0, 0), "constant", acts[2], 0.1, bnp, threads=8)
    assert got.shape == want.shape == (N, H, W, Co)
    np.testing.assert_allclose(got, want, err_msg=d, **TOL)
    np.testing.assert_allclose(got, pp(pd(ps(xt))).numpy(), err_msg=d, rtol=2e-5, atol=2e-5)
    np.testing.assert_array_equal(plan(xt).numpy(), got)
    monkeypatch.setenv("SNNHIP_STEM_MARCH", "0")   # the stem as its own launch, the pair still marching
    two = snn.chain_plan(ctx, [ps, pd, pp])
    assert two.num_steps() == 2 and "dwpw_march_f32" in two.describe() and "stem_dwpw" not in two.describe(), two.describe()
    np.testing.assert_allclose(two(xt).numpy(), got, err_msg=d, rtol=2e-5, atol=2e-5)


# N, H, W, C, Ch, Co, stride, residual, (act1, act2, act3): the whole-image kernel (irb_image_kernel; SNNHIP_IRB_IMAGE=1 takes it at any batch size)
IMAGE_CASES = [(2, 14, 14, 64, 384, 64, 1, True, ("relu6", "relu6", "")),      # MobileNetV2 b07-b09: 13 pixel tiles, 4 output blocks
               (2, 14, 14, 64, 384, 96, 1, False, ("relu6", "relu6", "")),     # b10
               (3, 14, 14, 96, 576, 96, 1, True, ("relu6", "relu6", "")),      # b11 / b12: 312 accumulator registers per lane
               (2, 14, 14, 96, 576, 160, 2, False, ("relu6", "relu6", "")),    # b13: 14x14 -> 7x7
               (3, 7, 7, 160, 960, 160, 1, True, ("relu6", "relu6", "")),      # b14 / b15
               (2, 10, 13, 64, 128, 64, 1, True, ("relu6", "relu6", "relu")),  # 130 pixels = 8 tiles + 2 pixels, two slices per wave, activations after project and add
               (1, 13, 11, 96, 192, 160, 2, False, ("relu6", "relu6", "leakyRelu")),  # odd extents at stride 2 (7x6 outputs = 3 tiles of the 4 the kernel walks)
               (2, 5, 9, 160, 64, 160, 1, False, ("relu6", "relu6", "")),      # one slice per wave
               (3, 7, 7, 160, 960, 320, 1, False, ("relu6", "relu6", ""))]     # b16: 20 output blocks = two blocks per image with ten each (round 6)


@pytest.mark.parametrize("case", IMAGE_CASES, ids=lambda c: "%dx%dx%d_%d-%d-%d_s%d%s" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], "_res" if c[7] else ""))
def test_irb_whole_image_kernel_matches_oracle_and_separate_layers(ctx, case, monkeypatch):
    """irb_image_kernel: block = one image, wave = a quarter of the hidden channels, partial sums reduced through LDS.  Against the CPU oracle layer by
    layer and against the separate HIP layers."""
    import shadernn_amd as snn

    monkeypatch.delenv("SNNHIP_IRB_FUSION", raising=False)
    monkeypatch.setenv("SNNHIP_IRB_IMAGE", "1")
    N, H, W, C, Ch, Co, s, res, acts = case
    x = _rand((N, H, W, C), 171)
    L = _layers(case, 180)
    (we, be, bne), (wd, bd, bnd), (wp, bp, bnp) = L
    pe = snn.conv2d_plan(ctx, N, H, W, we, be, act=acts[0], leaky=0.1, bn=bne)
    pd = snn.conv2d_plan(ctx, N, H, W, wd, bd, stride=s, pads=O.padding_offsets("same", 3), act=acts[1], leaky=0.1, bn=bnd, depthwise=True)
    _, OH, OW, _ = pd.out_shape()
    pp = snn.conv2d_plan(ctx, N, OH, OW, wp, bp, act=acts[2], leaky=0.1, bn=bnp)
    xt = snn.Tensor.from_numpy(ctx, x)
    sep = pp(pd(pe(xt)))
    want = _oracle(case, x, L)
    if res:
        act4 = "relu" if acts[2] == "relu" else ""
        want = O.add_act(O.conv2d(O.depthwise(O.conv2d(x, we, be, 1, (0, 0, 0, 0), "constant", acts[0], 0.1, bne, threads=8), wd, bd, s, O.padding_offsets("same", 3),
                                              acts[1], 0.1, bnd), wp, bp, 1, (0, 0, 0, 0), "constant", acts[2], 0.1, bnp, threads=8), x, act4)
        pa = snn.add_plan(ctx, N, OH, OW, Co, act=act4)
        sep = pa([sep, xt])
        fused = snn.graph_fuse(ctx, [(pe, [-1], False), (pd, [0], False), (pp, [1], False), (pa, [2, -1], True)])
        assert [f[0] is None for f in fused] == [True, True, True, False], [f[0] and f[0].describe() for f in fused]
        plan, ins = fused[3]
        assert ins == [-1]
    else:
        plan = snn.chain_plan(ctx, [pe, pd, pp])
        assert plan.num_steps() == 1
    d = plan.describe()
    assert "irb_fused" in d and "image per block" in d and "irb_image_kernel" in d and ("+ add" in d) == res, d
    got = plan(xt).numpy()
    assert got.shape == want.shape == (N, OH, OW, Co)
    np.testing.assert_allclose(got, want, err_msg=d, **TOL)
    np.testing.assert_allclose(got, sep.numpy(), err_msg=d, rtol=2e-5, atol=2e-5)
    np.testing.assert_array_equal(got, plan(xt).numpy())  # fixed reduction order
    f, b = plan.cost()
    assert f > 0 and b > 0


def test_irb_whole_image_kernel_is_the_default_on_small_maps_at_large_batch_only(ctx, monkeypatch):
    import shadernn_amd as snn

    monkeypatch.delenv("SNNHIP_IRB_FUSION", raising=False)
    monkeypatch.delenv("SNNHIP_IRB_IMAGE", raising=False)

    def steps(N, H, W, C, Ch, Co, s):
        case = (N, H, W, C, Ch, Co, s, False, ("relu6", "relu6", ""))
        (we, be, bne), (wd, bd, bnd), (wp, bp, bnp) = _layers(case, 5)
        pe = snn.conv2d_plan(ctx, N, H, W, we, be, act="relu6", bn=bne)
        pd = snn.conv2d_plan(ctx, N, H, W, wd, bd, stride=s, pads=O.padding_offsets("same", 3), act="relu6", bn=bnd, depthwise=True)
        _, OH, OW, _ = pd.out_shape()
        pp = snn.conv2d_plan(ctx, N, OH, OW, wp, bp, bn=bnp)
        try:
            plan = snn.chain_plan(ctx, [pe, pd, pp])
        except snn.SnnHipError as e:  # no rule takes the three layers: they stay three launches
            return 3, str(e)
        return plan.num_steps(), plan.describe()

    n, d = steps(256, 14, 14, 64, 384, 64, 1)
    assert n == 1 and "image per block" in d, d
    n, d = steps(8, 14, 14, 64, 384, 64, 1)      # eight images would leave 248 CUs idle: the three layers
    assert n == 3, d
    n, d = steps(256, 14, 14, 64, 384, 72, 1)    # not whole 16-channel output blocks
    assert n == 3, d
    monkeypatch.setenv("SNNHIP_IRB_IMAGE", "0")
    n, d = steps(256, 14, 14, 64, 384, 64, 1)
    assert n == 3, d


# N, H, W, C, Ch, Co, stride, residual, (act1, act2, act3): the band kernel (irb_band_kernel; SNNHIP_IRB_BAND=1 takes it at any batch size)
BAND_CASES = [((2, 112, 112, 16, 96, 24, 2, False, ("relu6", "relu6", "")), None),       # MobileNetV2 b01: column strips, stride 2
              ((2, 56, 56, 24, 144, 24, 1, True, ("relu6", "relu6", "")), None),        # b02: 24 channels = one full 16-channel step + the 8-channel tail
              ((2, 56, 56, 24, 144, 32, 2, False, ("relu6", "relu6", "")), None),       # b03
              ((3, 28, 28, 32, 192, 32, 1, True, ("relu6", "relu6", "")), None),        # b04 / b05
              ((2, 28, 28, 32, 192, 64, 2, False, ("relu6", "relu6", "")), None),       # b06: four output blocks
              ((1, 45, 37, 32, 80, 32, 1, True, ("relu6", "relu6", "relu")), "4,20,5"),  # ragged: last band 1 row, last strip 17 columns, 5 waves
              ((2, 33, 50, 8, 48, 24, 2, False, ("relu6", "relu6", "leakyRelu")), "3,13,4"),  # C = 8: only the tail step; 17x25 outputs in bands of 3 x 13
              ((1, 30, 30, 32, 64, 64, 1, False, ("relu6", "relu6", "")), "7,30,8"),     # four output blocks, 8 waves
              ((2, 56, 56, 24, 144, 24, 1, True, ("relu6", "relu6", "")), "8,28,8"),     # b02 at the geometry batch 256 pins (round 6: eight waves, tools/r6_geom.sh)
              ((2, 56, 56, 24, 144, 32, 2, False, ("relu6", "relu6", "")), "4,28,8"),    # b03 at its pinned geometry: stride 2 with eight waves
              ((3, 28, 28, 32, 192, 32, 1, True, ("relu6", "relu6", "")), "7,28,7")]     # b04 / b05 at the geometry batch 256 pins for the split-precision kernel


@pytest.mark.parametrize("case,geom", BAND_CASES, ids=lambda c: ("%dx%dx%d_%d-%d-%d_s%d%s" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], "_res" if c[7] else "")) if isinstance(c, tuple) else str(c))
def test_irb_band_kernel_matches_oracle_and_separate_layers(ctx, case, geom, monkeypatch):
    """irb_band_kernel: block = a band of output rows x a column strip of one image, the waves share the expand layer's pixel tiles and own their output
    tiles, two hidden buffers and one barrier per slice.  Against the CPU oracle layer by layer and against the separate HIP layers."""
    import shadernn_amd as snn

    monkeypatch.delenv("SNNHIP_IRB_FUSION", raising=False)
    monkeypatch.setenv("SNNHIP_IRB_BAND", "1")
    if geom:
        monkeypatch.setenv("SNNHIP_IRB_BAND_GEOM", geom)
    N, H, W, C, Ch, Co, s, res, acts = case
    x = _rand((N, H, W, C), 271)
    L = _layers(case, 280)
    (we, be, bne), (wd, bd, bnd), (wp, bp, bnp) = L
    pe = snn.conv2d_plan(ctx, N, H, W, we, be, act=acts[0], leaky=0.1, bn=bne)
    pd = snn.conv2d_plan(ctx, N, H, W, wd, bd, stride=s, pads=O.padding_offsets("same", 3), act=acts[1], leaky=0.1, bn=bnd, depthwise=True)
    _, OH, OW, _ = pd.out_shape()
    pp = snn.conv2d_plan(ctx, N, OH, OW, wp, bp, act=acts[2], leaky=0.1, bn=bnp)
    xt = snn.Tensor.from_numpy(ctx, x)
    sep = pp(pd(pe(xt)))
    h = O.conv2d(x, we, be, 1, (0, 0, 0, 0), "constant", acts[0], 0.1, bne, threads=8)
    dd = O.depthwise(h, wd, bd, s, O.padding_offsets("same", 3), acts[1], 0.1, bnd)
    want = O.conv2d(dd, wp, bp, 1, (0, 0, 0, 0), "constant", acts[2], 0.1, bnp, threads=8)
    if res:
        act4 = "relu" if acts[2] == "relu" else ""
        want = O.add_act(want, x, act4)
        pa = snn.add_plan(ctx, N, OH, OW, Co, act=act4)
        sep = pa([sep, xt])
        fused = snn.graph_fuse(ctx, [(pe, [-1], False), (pd, [0], False), (pp, [1], False), (pa, [2, -1], True)])
        assert [f[0] is None for f in fused] == [True, True, True, False], [f[0] and f[0].describe() for f in fused]
        plan, ins = fused[3]
        assert ins == [-1]
    else:
        plan = snn.chain_plan(ctx, [pe, pd, pp])
        assert plan.num_steps() == 1
    d = plan.describe()
    assert "irb_fused" in d and "band per block" in d and "irb_band_kernel" in d and ("+ add" in d) == res and ("tail8" in d) == (C % 16 == 8), d
    got = plan(xt).numpy()
    assert got.shape == want.shape == (N, OH, OW, Co)
    np.testing.assert_allclose(got, want, err_msg=d, **TOL)
    np.testing.assert_allclose(got, sep.numpy(), err_msg=d, rtol=2e-5, atol=2e-5)
    np.testing.assert_array_equal(got, plan(xt).numpy())


# ---- the split-precision form of the pointwise stages (round 6): fp32 operands as fp16 hi + lo, three f16 MFMA products per fp32 product.  The cases name the kernel
# (environment that selects it), the block shape and whether the block input / the expand weights sit outside fp16's range
SPLIT_KERNELS = {"image": ({"SNNHIP_IRB_IMAGE": "1"}, "irb_image_kernel", (2, 14, 14, 64, 384, 64, 1, True, ("relu6", "relu6", ""))),
                 "image96": ({"SNNHIP_IRB_IMAGE": "1"}, "irb_image_kernel", (2, 14, 14, 96, 576, 96, 1, True, ("relu6", "relu6", ""))),
                 "band": ({"SNNHIP_IRB_BAND": "1"}, "irb_band_kernel", (2, 56, 56, 24, 144, 24, 1, True, ("relu6", "relu6", ""))),
                 "band_s2": ({"SNNHIP_IRB_BAND": "1"}, "irb_band_kernel", (2, 28, 28, 32, 192, 64, 2, False, ("relu6", "relu6", ""))),
                 "wave": ({"SNNHIP_IRB_BAND": "0"}, "irb_wave_kernel", (2, 112, 112, 16, 96, 24, 2, False, ("relu6", "relu6", ""))),
                 "wave_res": ({"SNNHIP_IRB_BAND": "0"}, "irb_wave_kernel", (2, 56, 56, 24, 144, 24, 1, True, ("relu6", "relu6", "")))}


def _split_block(ctx, monkeypatch, which, split, x_scale=1.0, we_scale=1.0, wp_scale=1.0):
    import shadernn_amd as snn

    env, kernel, case = SPLIT_KERNELS[which]
    monkeypatch.delenv("SNNHIP_IRB_FUSION", raising=False)
    for k in ("SNNHIP_IRB_IMAGE", "SNNHIP_IRB_BAND", "SNNHIP_IRB_BAND_GEOM"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    if split is None:
        monkeypatch.delenv("SNNHIP_IRB_SPLIT", raising=False)
    else:
        monkeypatch.setenv("SNNHIP_IRB_SPLIT", split)
    N, H, W, C, Ch, Co, s, res, acts = case
    x = _rand((N, H, W, C), 371) * np.float32(x_scale)
    (we, be, bne), (wd, bd, bnd), (wp, bp, bnp) = _layers(case, 380)
    we, wp = we * np.float32(we_scale), wp * np.float32(wp_scale)
    pe = snn.conv2d_plan(ctx, N, H, W, we, be, act=acts[0], leaky=0.1, bn=bne)
    pd = snn.conv2d_plan(ctx, N, H, W, wd, bd, stride=s, pads=O.padding_offsets("same", 3), act=acts[1], leaky=0.1, bn=bnd, depthwise=True)
    _, OH, OW, _ = pd.out_shape()
    pp = snn.conv2d_plan(ctx, N, OH, OW, wp, bp, act=acts[2], leaky=0.1, bn=bnp)
    xt = snn.Tensor.from_numpy(ctx, x)
    h = O.conv2d(x, we, be, 1, (0, 0, 0, 0), "constant", acts[0], 0.1, bne, threads=8)
    dd = O.depthwise(h, wd, bd, s, O.padding_offsets("same", 3), acts[1], 0.1, bnd)
    want = O.conv2d(dd, wp, bp, 1, (0, 0, 0, 0), "constant", acts[2], 0.1, bnp, threads=8)
    if res:
        want = O.add_act(want, x, "")
        pa = snn.add_plan(ctx, N, OH, OW, Co, act="")
        plan, ins = snn.graph_fuse(ctx, [(pe, [-1], False), (pd, [0], False), (pp, [1], False), (pa, [2, -1], True)])[3]
        assert ins == [-1]
    else:
        plan = snn.chain_plan(ctx, [pe, pd, pp])
        assert plan.num_steps() == 1
    d = plan.describe()
    assert kernel in d, d
    return plan(xt).numpy(), want, d


@pytest.mark.parametrize("which", sorted(SPLIT_KERNELS))
def test_irb_split_form_is_the_default_and_stays_within_1e5_of_the_fp32_mfma_form(ctx, which, monkeypatch):
    """Default = the split form (description says so); SNNHIP_IRB_SPLIT=0 = the fp32 MFMA form of rounds 4-5.  Both against the oracle at the north-star
    tolerance, and against each other an order of magnitude tighter: the split products carry 22 bits."""
    got, want, d = _split_block(ctx, monkeypatch, which, None)
    assert "f16x3split" in d and ",true>" in d.split("kernel=")[1], d
    ref, _, d0 = _split_block(ctx, monkeypatch, which, "0")
    assert "mfma_f32_16x16x4" in d0 and "f16x3split" not in d0, d0
    np.testing.assert_allclose(got, want, err_msg=d, **TOL)
    np.testing.assert_allclose(ref, want, err_msg=d0, **TOL)
    np.testing.assert_allclose(got, ref, err_msg=d, rtol=1e-5, atol=1e-5)
    one, _, _ = _split_block(ctx, monkeypatch, which, "1")
    np.testing.assert_array_equal(one, got)


@pytest.mark.parametrize("which", sorted(SPLIT_KERNELS))
def test_irb_split_form_block_inputs_and_weight_rows_outside_the_fp16_range(ctx, which, monkeypatch):
    """A block input of magnitude 4e5 (fp16 ends at 65504: the kernel scales the tile by a power of two and undoes it after the expand MFMAs) against expand
    weights of 1e-6 (fp16's normal range starts at 6e-5: the host multiplies every weight row by a power of two and folds its inverse into the epilogue /
    the depthwise taps), and project weights of 3e3 x the usual.  Same tolerance as every other block test, relative to the output's own scale."""
    got, want, d = _split_block(ctx, monkeypatch, which, None, x_scale=1e5, we_scale=1e-5, wp_scale=1.0)
    assert "f16x3split" in d, d
    np.testing.assert_allclose(got, want, err_msg=d, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(want).max()) / 10.0))
    got, want, d = _split_block(ctx, monkeypatch, which, None, x_scale=1e-3, we_scale=1e3, wp_scale=3e3)
    scale = float(np.abs(want).max())
    np.testing.assert_allclose(got, want, err_msg=d, rtol=1e-4, atol=1e-5 * scale)
