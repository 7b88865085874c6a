"""conv2d_wino.hip: fp32 3x3 stride-1 convolutions as Winograd F(2x2,3x3) on the matrix pipe -- against the CPU oracle (north-star tolerance),
against the direct implicit-GEMM kernel on the same inputs, with every epilogue (bias / BN / activations / fused residual Add), ragged
extents, batch tiles, the split-K path and the reference's "valid" size rule (Q20)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from test_ops_gpu import _bn, _rand

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-4)


def _plan(ctx, family, *a, **kw):
    import shadernn_amd as snn

    os.environ["SNNHIP_CONV"] = family
    try:
        return snn.conv2d_plan(ctx, *a, **kw)
    finally:
        os.environ.pop("SNNHIP_CONV", None)


# N, H, W, IC, OC, act, bn, pads
CASES = [(2, 56, 56, 64, 64, "relu", True, (1, 1, 1, 1)),      # ResNet-18 layer1 shape: 4x64 px tiles
         (3, 28, 28, 128, 128, "relu", True, (1, 1, 1, 1)),    # two images per block tile
         (2, 14, 14, 256, 256, "", True, (1, 1, 1, 1)),        # 7x7 Winograd tiles per image, split-K
         (5, 7, 7, 512, 512, "relu", False, (1, 1, 1, 1)),     # four images per block tile (last one ragged), split-K
         (2, 17, 23, 32, 48, "leakyRelu", True, (1, 1, 1, 1)), # odd extents, OC not a multiple of 64
         (1, 9, 9, 40, 16, "tanh", False, (1, 1, 1, 1)),       # IC % 16 == 8, one 16-channel output block, non-simple activation
         (1, 33, 65, 8, 32, "relu6", False, (1, 1, 1, 1)),     # a single 8-channel chunk
         (2, 12, 20, 64, 96, "sigmoid", True, (0, 0, 0, 0)),   # "valid": pads 0 and the unshrunk output extent of the reference's size rule (Q20)
         (1, 64, 64, 32, 32, "SiLU", False, (1, 1, 1, 1)),
         (1, 6, 130, 48, 80, "", False, (1, 1, 1, 1))]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c[:5]) + "_" + (c[5] or "linear"))
def test_wino_matches_oracle_and_direct_kernel(ctx, case):
    import shadernn_amd as snn

    N, H, W, IC, OC, act, use_bn, pads = case
    x = _rand((N, H, W, IC), 31)
    w = _rand((OC, IC, 3, 3), 32, 1.0 / np.sqrt(9 * IC))
    b = _rand((OC,), 33, 0.1)
    bn = _bn(OC, 34) if use_bn else None
    pw = _plan(ctx, "wino", N, H, W, w, b, stride=1, pads=pads, act=act, leaky=0.1, bn=bn)
    assert "wino" in pw.describe(), pw.describe()
    xt = snn.Tensor.from_numpy(ctx, x)
    got = pw(xt).numpy()
    want = O.conv2d(x, w, b, 1, pads, "constant", act, 0.1, bn, threads=8)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, err_msg=pw.describe(), **TOL)
    pd = _plan(ctx, "mfma", N, H, W, w, b, stride=1, pads=pads, act=act, leaky=0.1, bn=bn)
    assert "wino" not in pd.describe()
    np.testing.assert_allclose(got, pd(xt).numpy(), err_msg=pw.describe() + " vs " + pd.describe(), rtol=2e-5, atol=2e-5)


def test_wino_is_the_default_for_gemm_sized_3x3_layers(ctx):
    import shadernn_amd as snn

    w = _rand((64, 64, 3, 3), 1, 0.05)
    assert "wino" in snn.conv2d_plan(ctx, 2, 20, 20, w, act="relu").describe()
    assert "wino" not in snn.conv2d_plan(ctx, 2, 20, 20, w, stride=2, act="relu").describe()                       # stride 2
    assert "wino" not in snn.conv2d_plan(ctx, 2, 20, 20, w, pad_mode="reflect").describe()                         # reflect padding
    assert "wino" not in snn.conv2d_plan(ctx, 2, 20, 20, _rand((64, 16, 3, 3), 2, 0.05)).describe()               # thin input: direct kernel
    assert "wino" not in snn.conv2d_plan(ctx, 2, 20, 20, _rand((64, 64, 5, 5), 3, 0.05)).describe()               # 5x5
    os.environ["SNNHIP_CONV_WINO"] = "0"
    try:
        assert "wino" not in snn.conv2d_plan(ctx, 2, 20, 20, w, act="relu").describe()
    finally:
        os.environ.pop("SNNHIP_CONV_WINO")
    assert "wino" not in snn.conv2d_plan(ctx, 2, 20, 20, w, act="relu", dtype=snn.F16).describe()                  # fp16 stays on the fp16 MFMA kernel


@pytest.mark.parametrize("shape", [(4, 28, 28, 128), (8, 7, 7, 512), (2, 19, 21, 64)], ids=["28x28x128", "7x7x512_splitK", "19x21x64"])
def test_wino_fused_residual_add(ctx, shape):
    """Chain rule E on the Winograd kernel: conv -> Add(+relu) as one launch (split-K: the reduce pass adds the residual)."""
    import shadernn_amd as snn

    N, H, W, C = shape
    x, res = _rand((N, H, W, C), 41), _rand((N, H, W, C), 42)
    w, b, bn = _rand((C, C, 3, 3), 43, 1.0 / np.sqrt(9 * C)), _rand((C,), 44, 0.1), _bn(C, 45)
    conv = snn.conv2d_plan(ctx, N, H, W, w, b, act="", bn=bn)
    add = snn.add_plan(ctx, N, H, W, C, act="relu")
    fused = snn.chain_plan(ctx, [conv, add])
    assert "wino" in fused.describe() and "+add" in fused.describe() and fused.num_steps() == 1, fused.describe()
    got = fused([snn.Tensor.from_numpy(ctx, x), snn.Tensor.from_numpy(ctx, res)]).numpy()
    want = O.add_act(O.conv2d(x, w, b, 1, (1, 1, 1, 1), "constant", "", 0.0, bn, threads=8), res, "relu")
    np.testing.assert_allclose(got, want, err_msg=fused.describe(), **TOL)


def test_wino_resnet_shapes_batch32_properties(ctx):
    """The four ResNet-18 body shapes at the benchmark batch: every batch position of a replicated image returns the same tensor, bit for bit,
    and image 0 matches the oracle."""
    import shadernn_amd as snn

    for H, C in ((56, 64), (28, 128), (14, 256), (7, 512)):
        x1 = _rand((1, H, H, C), 50 + H)
        w, b = _rand((C, C, 3, 3), 51, 1.0 / np.sqrt(9 * C)), _rand((C,), 52, 0.1)
        p = snn.conv2d_plan(ctx, 32, H, H, w, b, act="relu")
        assert "wino" in p.describe()
        got = p(snn.Tensor.from_numpy(ctx, np.repeat(x1, 32, axis=0))).numpy()
        for i in (1, 13, 31):
            np.testing.assert_array_equal(got[i], got[0], err_msg="%s image %d" % (p.describe(), i))
        np.testing.assert_allclose(got[:1], O.conv2d(x1, w, b, 1, (1, 1, 1, 1), "constant", "relu", threads=8), err_msg=p.describe(), **TOL)


def _plan_kgroups(ctx, kgroups, *a, **kw):
    import shadernn_amd as snn

    os.environ["SNNHIP_WINO_KGROUPS"] = str(kgroups)
    try:
        return snn.conv2d_plan(ctx, *a, **kw)
    finally:
        os.environ.pop("SNNHIP_WINO_KGROUPS")


# N, H, W, C_in, C_out, act, bn
KGROUP_CASES = [(32, 14, 14, 256, 256, "relu", True),    # ResNet-18 stage 3 at the benchmark batch: split-K 2 + reduce launch -> one launch
                (32, 7, 7, 512, 512, "relu", True),      # stage 4: split-K 4 -> two K groups x split-K 2
                (5, 7, 7, 512, 512, "", False),          # a ragged image group
                (3, 28, 28, 128, 128, "relu", True),     # forced on a layer the planner does not split
                (2, 17, 23, 32, 48, "leakyRelu", True),  # four chunks: two per group; odd extents, a half-empty channel block
                (1, 9, 9, 16, 16, "tanh", False)]        # two chunks: ONE per group; a single 16-channel output block (group 1's block is beyond OC)


@pytest.mark.parametrize("case", KGROUP_CASES, ids=lambda c: "x".join(str(v) for v in c[:5]) + "_" + (c[5] or "linear"))
def test_wino_k_groups_inside_the_block_match_the_split_over_blocks(ctx, case):
    """The channel chunks split over two groups of four waves INSIDE a block (sums exchanged through LDS, no reduce launch for the first factor of two)
    against the split over blockIdx.z + reduce pass on the same inputs, and against the oracle."""
    import shadernn_amd as snn

    N, H, W, IC, OC, act, use_bn = case
    x = _rand((N, H, W, IC), 131)
    w, b = _rand((OC, IC, 3, 3), 132, 1.0 / np.sqrt(9 * IC)), _rand((OC,), 133, 0.1)
    bn = _bn(OC, 134) if use_bn else None
    xt = snn.Tensor.from_numpy(ctx, x)
    os.environ["SNNHIP_CONV"] = "wino"
    try:
        p2 = _plan_kgroups(ctx, 2, N, H, W, w, b, act=act, leaky=0.1, bn=bn)
        p1 = _plan_kgroups(ctx, 1, N, H, W, w, b, act=act, leaky=0.1, bn=bn)
    finally:
        os.environ.pop("SNNHIP_CONV")
    assert "kgroups=2" in p2.describe() and "kgroups=1" in p1.describe(), (p2.describe(), p1.describe())
    got = p2(xt).numpy()
    np.testing.assert_array_equal(got, p2(xt).numpy(), err_msg="second run differs: " + p2.describe())
    np.testing.assert_allclose(got, p1(xt).numpy(), err_msg=p2.describe() + " vs " + p1.describe(), rtol=2e-5, atol=2e-5)
    n = min(N, 2)
    np.testing.assert_allclose(got[:n], O.conv2d(x[:n], w, b, 1, (1, 1, 1, 1), "constant", act, 0.1, bn, threads=8), err_msg=p2.describe(), **TOL)
    if N > 2:
        np.testing.assert_allclose(got[-1:], O.conv2d(x[-1:], w, b, 1, (1, 1, 1, 1), "constant", act, 0.1, bn, threads=8), err_msg=p2.describe(), **TOL)


def test_wino_k_groups_are_the_default_where_the_planner_splits(ctx):
    import shadernn_amd as snn

    d3 = snn.conv2d_plan(ctx, 32, 14, 14, _rand((256, 256, 3, 3), 1, 0.02), act="relu").describe()
    assert "kgroups=2 splitK=1" in d3, d3                                # no reduce launch left
    d4 = snn.conv2d_plan(ctx, 32, 7, 7, _rand((512, 512, 3, 3), 2, 0.02), act="relu").describe()
    assert "kgroups=1 splitK=4" in d4, d4                                # a deeper split keeps its reduce launch: the groups would not pay
    d1 = snn.conv2d_plan(ctx, 32, 56, 56, _rand((64, 64, 3, 3), 3, 0.05), act="relu").describe()
    assert "kgroups=1 splitK=1" in d1, d1                                # enough block tiles: nothing to split


@pytest.mark.parametrize("shape", [(32, 14, 14, 256), (6, 7, 7, 512)], ids=["14x14x256_b32", "7x7x512_b6"])
def test_wino_k_groups_with_the_fused_residual_add(ctx, shape):
    """Chain rule E with two K groups: at 14x14 the residual is added in the kernel's own epilogue (no reduce pass any more), at 7x7 in the reduce pass."""
    import shadernn_amd as snn

    N, H, W, C = shape
    x, res = _rand((N, H, W, C), 141), _rand((N, H, W, C), 142)
    w, b, bn = _rand((C, C, 3, 3), 143, 1.0 / np.sqrt(9 * C)), _rand((C,), 144, 0.1), _bn(C, 145)
    os.environ["SNNHIP_WINO_KGROUPS"] = "2"   # (the planner's own choice at 14x14; forced at 7x7)
    try:
        conv = snn.conv2d_plan(ctx, N, H, W, w, b, act="", bn=bn)
        fused = snn.chain_plan(ctx, [conv, snn.add_plan(ctx, N, H, W, C, act="relu")])
    finally:
        os.environ.pop("SNNHIP_WINO_KGROUPS")
    assert "kgroups=2" in fused.describe() and "+add" in fused.describe() and fused.num_steps() == 1, fused.describe()
    got = fused([snn.Tensor.from_numpy(ctx, x), snn.Tensor.from_numpy(ctx, res)]).numpy()
    for i in (0, N - 1):
        want = O.add_act(O.conv2d(x[i:i + 1], w, b, 1, (1, 1, 1, 1), "constant", "", 0.0, bn, threads=8), res[i:i + 1], "relu")
        np.testing.assert_allclose(got[i:i + 1], want, err_msg="%s image %d" % (fused.describe(), i), **TOL)
