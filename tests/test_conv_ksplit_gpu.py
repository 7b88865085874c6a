"""conv2d_ksplit.hip: fp32 convolutions as an implicit GEMM whose K axis is split over the waves of a block (operands straight from memory, partial
tiles summed through LDS; the 3x3 stride-2 layers of ResNet-18) -- against the CPU oracle (north-star tolerance), against the split-K + reduce
kernel it replaces on the same inputs, over every (tile, K-split, look-ahead) instantiation, ragged extents and last tiles, every epilogue, and at
the benchmark batch."""
import os

import numpy as np
import pytest

import oracle_lib as O
from test_ops_gpu import _bn, _rand

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-4)


def _plan(ctx, family, *a, geom=None, **kw):
    import shadernn_amd as snn

    if family:
        os.environ["SNNHIP_CONV"] = family
    if geom:
        os.environ["SNNHIP_KSPLIT"] = geom
    try:
        return snn.conv2d_plan(ctx, *a, **kw)
    finally:
        os.environ.pop("SNNHIP_CONV", None)
        os.environ.pop("SNNHIP_KSPLIT", None)


# N, H, W, IC, OC, k, stride, act, bn, pads
CASES = [(2, 56, 56, 64, 128, 3, 2, "relu", True, (1, 1, 1, 1)),       # ResNet-18 layer2 entry
         (3, 28, 28, 128, 256, 3, 2, "relu", True, (1, 1, 1, 1)),      # layer3 entry: tiles straddle images (196 pixels per image)
         (5, 14, 14, 256, 512, 3, 2, "", True, (1, 1, 1, 1)),          # layer4 entry: 49 pixels per image, a ragged last tile
         (2, 17, 23, 32, 64, 3, 2, "leakyRelu", True, (1, 1, 1, 1)),   # odd extents: the bottom / right taps leave the image too
         (1, 9, 9, 48, 64, 3, 2, "tanh", False, (1, 1, 1, 1)),         # three 16-channel chunks per tap, non-simple activation, M = 25 < one tile
         (2, 12, 20, 64, 128, 3, 2, "sigmoid", True, (0, 0, 0, 0)),    # "valid": pads 0 and the reference's size rule (Q20)
         (1, 33, 65, 16, 64, 3, 2, "relu6", False, (1, 1, 1, 1)),      # a single chunk per tap
         (2, 12, 20, 64, 64, 3, 1, "relu", True, (1, 1, 1, 1)),        # forced: stride 1
         (1, 15, 15, 16, 64, 5, 2, "SiLU", False, (2, 2, 2, 2)),       # forced: 5x5 stride 2
         (2, 10, 14, 32, 192, 1, 1, "", False, (0, 0, 0, 0)),          # forced: 1x1 (T = 2 iterations: fewer than some K splits have waves)
         (3, 56, 56, 64, 128, 1, 2, "", True, (0, 0, 0, 0)),           # ResNet-18's downsample branches (default plan): 1 / 2 / 4 waves per tile
         (3, 28, 28, 128, 256, 1, 2, "", True, (0, 0, 0, 0)),
         (5, 14, 14, 256, 512, 1, 2, "relu", True, (0, 0, 0, 0)),
         (2, 15, 17, 64, 64, 1, 2, "leakyRelu", False, (0, 0, 0, 0))]  # odd extents


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c[:7]) + "_" + (c[7] or "linear"))
def test_ksplit_matches_oracle_and_the_split_k_kernel(ctx, case):
    import shadernn_amd as snn

    N, H, W, IC, OC, k, s, act, use_bn, pads = case
    x = _rand((N, H, W, IC), 61)
    w = _rand((OC, IC, k, k), 62, 1.0 / np.sqrt(k * k * IC))
    b = _rand((OC,), 63, 0.1)
    bn = _bn(OC, 64) if use_bn else None
    pk = _plan(ctx, "ksplit", N, H, W, w, b, stride=s, pads=pads, act=act, leaky=0.1, bn=bn)
    assert "ksplit" in pk.describe(), pk.describe()
    xt = snn.Tensor.from_numpy(ctx, x)
    got = pk(xt).numpy()
    want = O.conv2d(x, w, b, s, pads, "constant", act, 0.1, bn, threads=8)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, err_msg=pk.describe(), **TOL)
    np.testing.assert_array_equal(got, pk(xt).numpy(), err_msg="second run differs: " + pk.describe())  # fixed summation order
    pd = _plan(ctx, "mfma", N, H, W, w, b, stride=s, pads=pads, act=act, leaky=0.1, bn=bn)
    assert "ksplit" not in pd.describe()
    np.testing.assert_allclose(got, pd(xt).numpy(), err_msg=pk.describe() + " vs " + pd.describe(), rtol=2e-5, atol=2e-5)


GEOMS = ["1,1,2", "1,2,2", "1,3,2", "1,4,2", "1,6,2", "1,8,2", "2,1,2", "2,2,2", "2,3,2", "2,4,2", "2,6,2", "2,8,2", "1,4,3", "2,4,3"]


@pytest.mark.parametrize("geom", GEOMS)
def test_every_instantiation_on_a_ragged_layer(ctx, geom):
    """(pixel tiles per wave, K split, look-ahead depth): each on a layer whose last pixel tile is partial and whose K iterations (27) divide into
    none of the splits evenly."""
    import shadernn_amd as snn

    N, H, W, IC, OC = 3, 13, 19, 48, 128
    x, w, b, bn = _rand((N, H, W, IC), 71), _rand((OC, IC, 3, 3), 72, 1.0 / np.sqrt(9 * IC)), _rand((OC,), 73, 0.1), _bn(OC, 74)
    p = _plan(ctx, "ksplit", N, H, W, w, b, stride=2, pads=(1, 1, 1, 1), act="relu", bn=bn, geom=geom)
    mi, ks, depth = (int(t) for t in geom.split(","))
    d = p.describe()
    assert "tile=%dpx" % (32 * mi) in d and "K over %d waves" % ks in d and "depth %d" % depth in d, d
    got = p(snn.Tensor.from_numpy(ctx, x)).numpy()
    np.testing.assert_allclose(got, O.conv2d(x, w, b, 2, (1, 1, 1, 1), "constant", "relu", 0.0, bn, threads=8), err_msg=d, **TOL)


def test_ksplit_is_the_default_for_fp32_3x3_stride_2_layers(ctx):
    import shadernn_amd as snn

    w = _rand((128, 64, 3, 3), 1, 0.05)
    assert "ksplit" in snn.conv2d_plan(ctx, 2, 20, 20, w, stride=2, act="relu").describe()
    assert "ksplit" not in snn.conv2d_plan(ctx, 2, 20, 20, w, stride=1, act="relu").describe()                        # stride 1: Winograd
    w1 = _rand((128, 64, 1, 1), 5, 0.1)
    assert "ksplit" in snn.conv2d_plan(ctx, 2, 20, 20, w1, stride=2, pads=(0, 0, 0, 0)).describe()                    # the 1x1 stride-2 sibling (downsample branch)
    assert "ksplit" not in snn.conv2d_plan(ctx, 2, 20, 20, w1, stride=1, pads=(0, 0, 0, 0)).describe()                # 1x1 stride 1: the stream kernel
    assert "ksplit" not in snn.conv2d_plan(ctx, 2, 20, 20, w, stride=2, pad_mode="reflect").describe()               # reflect padding
    assert "ksplit" not in snn.conv2d_plan(ctx, 2, 20, 20, _rand((64, 24, 3, 3), 2, 0.05), stride=2).describe()      # IC % 16 != 0
    assert "ksplit" not in snn.conv2d_plan(ctx, 2, 20, 20, _rand((96, 64, 3, 3), 3, 0.05), stride=2).describe()      # OC % 64 != 0
    assert "ksplit" not in snn.conv2d_plan(ctx, 2, 20, 20, _rand((64, 64, 5, 5), 4, 0.05), stride=2).describe()      # 5x5: only when forced
    assert "ksplit" not in snn.conv2d_plan(ctx, 2, 20, 20, w, stride=2, act="relu", dtype=snn.F16).describe()        # fp16 keeps its own kernels
    os.environ["SNNHIP_CONV_KSPLIT"] = "0"
    try:
        d = snn.conv2d_plan(ctx, 2, 20, 20, w, stride=2, act="relu").describe()
        assert "ksplit" not in d and "conv2d_mfma" in d, d
    finally:
        os.environ.pop("SNNHIP_CONV_KSPLIT")


def test_a_fused_add_behind_it_keeps_the_split_k_kernel(ctx):
    """Chain rule E is not offered by this kernel: the planner must fall back to conv2d_mfma's fused form, same result as the two launches."""
    import shadernn_amd as snn

    N, H, W, IC, OC = 2, 16, 16, 64, 128
    x, res = _rand((N, H, W, IC), 81), _rand((N, 8, 8, OC), 82)
    w, b = _rand((OC, IC, 3, 3), 83, 1.0 / np.sqrt(9 * IC)), _rand((OC,), 84, 0.1)
    conv = snn.conv2d_plan(ctx, N, H, W, w, b, stride=2, act="")
    assert "ksplit" in conv.describe()
    fused = snn.chain_plan(ctx, [conv, snn.add_plan(ctx, N, 8, 8, OC, act="relu")])
    got = fused([snn.Tensor.from_numpy(ctx, x), snn.Tensor.from_numpy(ctx, res)]).numpy()
    want = O.add_act(O.conv2d(x, w, b, 2, (1, 1, 1, 1), "constant", "", 0.0, None, threads=8), res, "relu")
    np.testing.assert_allclose(got, want, err_msg=fused.describe(), **TOL)


def test_ksplit_resnet_shapes_batch32_properties(ctx):
    """The three ResNet-18 stage entries at the benchmark batch through the default plan: every batch position of a replicated image returns the same
    tensor, bit for bit (a pixel's K order does not depend on where its tile sits), and image 0 matches the oracle."""
    import shadernn_amd as snn

    for H, C in ((56, 64), (28, 128), (14, 256)):
        x1 = _rand((1, H, H, C), 90 + H)
        w, b, bn = _rand((2 * C, C, 3, 3), 91, 1.0 / np.sqrt(9 * C)), _rand((2 * C,), 92, 0.1), _bn(2 * C, 93)
        p = snn.conv2d_plan(ctx, 32, H, H, w, b, stride=2, act="relu", bn=bn)
        assert "ksplit" in p.describe(), p.describe()
        got = p(snn.Tensor.from_numpy(ctx, np.repeat(x1, 32, axis=0))).numpy()
        for i in (1, 13, 31):
            np.testing.assert_array_equal(got[i], got[0], err_msg="%s image %d" % (p.describe(), i))
        np.testing.assert_allclose(got[:1], O.conv2d(x1, w, b, 2, (1, 1, 1, 1), "constant", "relu", 0.0, bn, threads=8), err_msg=p.describe(), **TOL)


def _fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        k = int(rng.choice([1, 2, 3, 3, 4, 5, 7]))
        s = int(rng.choice([1, 2, 2]))
        ic = int(rng.choice([16, 32, 48, 64, 96, 144]))
        oc = int(rng.choice([64, 128, 192]))
        h, w = int(rng.integers(max(k, 5), 36)), int(rng.integers(max(k, 5), 40))
        b = int(rng.choice([1, 2, 3, 5]))
        if b * h * w * ic * k * k * oc > 3e8:  # keep the oracle fast
            continue
        padding = "same" if k % 2 == 0 else str(rng.choice(["same", "valid"]))
        out.append((b, h, w, ic, oc, k, s, str(rng.choice(["", "relu", "relu6", "tanh", "sigmoid", "leakyRelu", "SiLU"])), bool(rng.integers(0, 2)), padding,
                    str(rng.choice(GEOMS)), int(rng.integers(0, 1 << 30))))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(40, 20261001), ids=lambda c: "n%d_%dx%d_%d-%d_k%ds%d_%s_bn%d_%s_g%s" % c[:11])
def test_ksplit_random_shapes_and_geometries_match_oracle(ctx, case):
    """Randomised sweep (fixed seed) of the kernel forced on shapes it is not the default for: kernel sizes 1 .. 7 (even ones with their asymmetric "same"
    padding), both strides, ragged maps and batches, every activation, a random (tile, K split, look-ahead) instantiation per case."""
    import shadernn_amd as snn

    n, h, w, ic, oc, k, s, act, use_bn, padding, geom, seed = case
    x = _rand((n, h, w, ic), seed)
    wt = _rand((oc, ic, k, k), seed + 1, 1.0 / np.sqrt(ic * k * k))
    b = _rand((oc,), seed + 2, 0.2)
    bn = _bn(oc, seed + 3) if use_bn else None
    pads = O.padding_offsets(padding, k)
    plan = _plan(ctx, "ksplit", n, h, w, wt, b, stride=s, pads=pads, act=act, leaky=0.15, bn=bn, geom=geom)
    assert "ksplit" in plan.describe(), plan.describe()
    got = plan(snn.Tensor.from_numpy(ctx, x)).numpy()
    want = O.conv2d(x, wt, b, s, pads, "constant", act, 0.15, bn)
    assert got.shape == want.shape, plan.describe()
    np.testing.assert_allclose(got, want, err_msg=plan.describe(), **TOL)


@pytest.mark.parametrize("case", [(4, 56, 56, 64, 128, 2), (4, 28, 28, 128, 256, 4), (5, 14, 14, 256, 512, 8), (3, 17, 23, 32, 64, 2), (2, 20, 20, 64, 64, 0)],
                         ids=lambda c: "x".join(map(str, c[:5])) + "_ks%d" % c[5])
def test_a_launch_group_runs_the_3x3_and_the_1x1_stride_2_layers_as_one_kernel(ctx, case):
    """snnhip_ctx_group_begin / _end around the two branches of a ResNet stage entry (same input): ONE launch (conv2d_ksplit_pair_kernel: the blocks of the
    1x1 problem behind the 3x3's; checked in the launch trace), bit-identical to the two launches at the same K split -- a tile's arithmetic does not
    depend on which grid it runs in -- and within the tolerance of the oracle; either order inside the group; a group of one and a group with a plan the
    library cannot defer behave like no group at all; SNNHIP_KSPLIT_NO_PAIRS=1 launches the two one by one at the group's end.  The geometries the bench
    batch gives the planner (64-pixel tiles, K over 2 / 4 / 8 waves) are pinned here at oracle-sized batches; ks0 = the planner's own choice at this size
    (32-pixel tiles: no pair kernel, the group launches one by one)."""
    import shadernn_amd as snn
    from shadernn_amd import capi

    N, H, W, IC, OC, ks = case
    x = _rand((N, H, W, IC), 101)
    w3, b3, bn3 = _rand((OC, IC, 3, 3), 102, 1.0 / np.sqrt(9 * IC)), _rand((OC,), 103, 0.1), _bn(OC, 104)
    w1, b1, bn1 = _rand((OC, IC, 1, 1), 105, 1.0 / np.sqrt(IC)), _rand((OC,), 106, 0.1), _bn(OC, 107)
    p3 = _plan(ctx, "", N, H, W, w3, b3, stride=2, pads=(1, 1, 1, 1), act="relu", bn=bn3, geom="2,%d,2" % ks if ks else None)
    p1 = _plan(ctx, "", N, H, W, w1, b1, stride=2, pads=(0, 0, 0, 0), act="", bn=bn1, geom="1,%d,2" % ks if ks else None)
    assert p3.groupable() and p1.groupable() and "ksplit" in p3.describe() and "ksplit" in p1.describe()
    xt = snn.Tensor.from_numpy(ctx, x)
    y3, y1 = p3(xt).numpy(), p1(xt).numpy()
    np.testing.assert_allclose(y3[:1], O.conv2d(x[:1], w3, b3, 2, (1, 1, 1, 1), "constant", "relu", 0.0, bn3, threads=8), **TOL)
    np.testing.assert_allclose(y1[:1], O.conv2d(x[:1], w1, b1, 2, (0, 0, 0, 0), "constant", "", 0.0, bn1, threads=8), **TOL)

    def grouped(first, second, also=None):
        outs = {id(p3): snn.Tensor(ctx, *p3.out_shape()), id(p1): snn.Tensor(ctx, *p1.out_shape())}
        for t in outs.values():
            t.fill(-7.0)
        capi.trace_begin()
        ctx.group_begin()
        first(xt, outs[id(first)])
        if also is not None:
            also()
        if second is not None:
            second(xt, outs[id(second)])
        ctx.group_end()
        ctx.sync()
        fns = sorted(k["function"] for k in capi.trace_end()["kernels"])
        return outs[id(p3)].numpy(), outs[id(p1)].numpy(), fns

    g3, g1, fns = grouped(p3, p1)
    assert fns == (["conv2d_ksplit_pair_kernel"] if ks else ["conv2d_ksplit_kernel"]), fns
    np.testing.assert_array_equal(g3, y3)
    np.testing.assert_array_equal(g1, y1)
    g3, g1, fns = grouped(p1, p3)                # the downsample first
    assert fns == (["conv2d_ksplit_pair_kernel"] if ks else ["conv2d_ksplit_kernel"]), fns
    np.testing.assert_array_equal(g3, y3)
    np.testing.assert_array_equal(g1, y1)
    g3, g1, fns = grouped(p3, None)              # a group of one
    assert fns == ["conv2d_ksplit_kernel"], fns
    np.testing.assert_array_equal(g3, y3)
    assert (g1 == -7.0).all()
    add = snn.add_plan(ctx, N, H, W, IC, act="relu")
    assert not add.groupable()
    side = {}
    g3, g1, fns = grouped(p3, p1, also=lambda: side.setdefault("y", add([xt, xt])))   # a plan that launches at once, between the two deferred ones
    assert len(fns) == 2 and ("conv2d_ksplit_pair_kernel" in fns) == bool(ks), fns
    np.testing.assert_array_equal(g3, y3)
    np.testing.assert_array_equal(g1, y1)
    np.testing.assert_array_equal(side["y"].numpy(), np.maximum(x + x, 0.0))
    os.environ["SNNHIP_KSPLIT_NO_PAIRS"] = "1"
    try:
        g3, g1, fns = grouped(p3, p1)
    finally:
        os.environ.pop("SNNHIP_KSPLIT_NO_PAIRS")
    assert fns == ["conv2d_ksplit_kernel"], fns
    np.testing.assert_array_equal(g3, y3)
    np.testing.assert_array_equal(g1, y1)
    with pytest.raises(snn.SnnHipError, match="no group is open"):
        ctx.group_end()
