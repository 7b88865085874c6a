"""GPU parity tests of the element-wise / pooling / shape operators through the C-ABI vs the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as O
from test_ops_gpu import TOL, _bn, _rand

pytestmark = pytest.mark.gpu
ACTS = ["", "relu", "relu6", "tanh", "sigmoid", "leakyRelu", "SiLU"]


def _run(ctx, plan, *xs):
    import shadernn_amd as snn

    ts = [snn.Tensor.from_numpy(ctx, x) for x in xs]
    yt = plan(ts if len(ts) > 1 else ts[0])
    y = yt.numpy()
    for t in ts + [yt]:
        t.free()
    desc = plan.describe()
    plan.destroy()
    return y, desc


@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("shape", [(2, 9, 11, 16), (1, 7, 5, 3)])
def test_add_matches_oracle(ctx, act, shape):
    import shadernn_amd as snn

    a, b = _rand(shape, 1), _rand(shape, 2)
    y, desc = _run(ctx, snn.add_plan(ctx, *shape, act=act, leaky=0.2), a, b)
    np.testing.assert_allclose(y, O.add_act(a, b, act, 0.2), err_msg=desc, **TOL)


@pytest.mark.parametrize("act", ACTS[1:])
def test_activation_matches_oracle(ctx, act):
    import shadernn_amd as snn

    a = _rand((2, 6, 7, 12), 3, 3.0)
    y, desc = _run(ctx, snn.activation_plan(ctx, 2, 6, 7, 12, act, 0.3), a)
    np.testing.assert_allclose(y, O.add_act(a, None, act, 0.3), err_msg=desc, **TOL)


@pytest.mark.parametrize("act", ["", "relu", "relu6"])
@pytest.mark.parametrize("shape", [(2, 5, 6, 24), (1, 4, 9, 7)])
def test_batchnorm_matches_oracle(ctx, act, shape):
    import shadernn_amd as snn

    x = _rand(shape, 4)
    bn = _bn(shape[3], 5)
    y, desc = _run(ctx, snn.batchnorm_plan(ctx, *shape, bn, act=act), x)
    np.testing.assert_allclose(y, O.batchnorm(x, bn, act), err_msg=desc, **TOL)


@pytest.mark.parametrize("kind", ["max", "avg"])
@pytest.mark.parametrize("n,h,w,c,k,s,same", [(2, 112, 112, 64, 3, 2, True),   # ResNet-18 stem pool
                                               (1, 9, 11, 8, 2, 2, False), (1, 9, 11, 8, 3, 2, False), (2, 8, 8, 5, 3, 1, True),
                                               (1, 7, 7, 12, 7, 7, False), (3, 5, 6, 4, 2, 1, True)])
def test_pool2d_matches_oracle(ctx, kind, n, h, w, c, k, s, same):
    import shadernn_amd as snn

    x = _rand((n, h, w, c), 6)
    y, desc = _run(ctx, snn.pool2d_plan(ctx, n, h, w, c, k, s, kind=kind, same=same), x)
    want = O.pool2d(x, k, s, kind, same)
    assert y.shape == want.shape, desc
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)


def test_global_avgpool_matches_oracle_and_mean(ctx):
    import shadernn_amd as snn

    x = _rand((4, 7, 7, 512), 7)
    y, desc = _run(ctx, snn.global_avgpool_plan(ctx, 4, 7, 7, 512), x)
    np.testing.assert_allclose(y, O.global_avgpool(x), err_msg=desc, **TOL)
    np.testing.assert_allclose(y[:, 0, 0, :], x.mean(axis=(1, 2)), rtol=1e-4, atol=1e-5)
    # the 16-lanes-per-output kernel: pixel counts that are not multiples of 16, more outputs than one block, half tensors; maps below 16 pixels
    # and channel counts that are not multiples of 4 stay on the general pooling kernel
    for shape in [(3, 5, 9, 1280), (33, 7, 7, 64), (2, 3, 3, 32), (2, 6, 6, 6)]:
        x = _rand(shape, 8)
        y, desc = _run(ctx, snn.global_avgpool_plan(ctx, *shape), x)
        np.testing.assert_allclose(y, O.global_avgpool(x), err_msg=str(shape), **TOL)
    x = _rand((2, 7, 7, 128), 9)
    p = snn.global_avgpool_plan(ctx, 2, 7, 7, 128)
    y16 = p(snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)).numpy()
    np.testing.assert_allclose(y16, O._h(O.global_avgpool(O._h(x))), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("mode", ["constant", "replicate", "reflect"])
@pytest.mark.parametrize("shape,pads", [((1, 10, 12, 3), (4, 4, 4, 4)), ((2, 6, 7, 8), (1, 2, 3, 0)), ((1, 5, 5, 4), (2, 2, 2, 2))])
def test_pad_matches_oracle(ctx, mode, shape, pads):
    import shadernn_amd as snn

    x = _rand(shape, 8)
    y, desc = _run(ctx, snn.pad_plan(ctx, *shape, pads, mode), x)
    want = O.pad(x, pads, mode)
    assert y.shape == want.shape
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)


@pytest.mark.parametrize("mode", ["nearest", "bilinear"])
@pytest.mark.parametrize("shape,scale", [((1, 9, 11, 8), 2.0), ((2, 5, 7, 3), 2.0), ((1, 6, 6, 4), 3.0), ((1, 8, 8, 4), 1.5)])
def test_upsample_matches_oracle(ctx, mode, shape, scale):
    import shadernn_amd as snn

    x = _rand(shape, 9)
    y, desc = _run(ctx, snn.upsample_plan(ctx, *shape, scale, mode), x)
    want = O.upsample(x, scale, mode)
    assert y.shape == want.shape
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)


@pytest.mark.parametrize("act", ["", "relu"])
@pytest.mark.parametrize("shape", [(2, 45, 80, 32), (1, 17, 23, 5), (3, 4, 4, 128), (1, 180, 320, 16)])
def test_instancenorm_matches_oracle(ctx, act, shape):
    import shadernn_amd as snn

    x = _rand(shape, 10, 2.0) + 0.7
    r = np.random.default_rng(11)
    beta, gamma = r.uniform(-0.5, 0.5, shape[3]).astype(np.float32), r.uniform(0.5, 1.5, shape[3]).astype(np.float32)
    y, desc = _run(ctx, snn.instancenorm_plan(ctx, *shape, beta, gamma, act=act), x)
    np.testing.assert_allclose(y, O.instancenorm(x, beta, gamma, act), err_msg=desc, **TOL)


def test_input_count_is_checked(ctx):
    import shadernn_amd as snn

    p = snn.add_plan(ctx, 1, 4, 4, 4)
    t = snn.Tensor.from_numpy(ctx, np.zeros((1, 4, 4, 4), np.float32))
    with pytest.raises(snn.SnnHipError):
        p(t)


@pytest.mark.parametrize("s0,s1", [((2, 9, 11, 8), (2, 5, 7, 8)), ((1, 5, 7, 3), (1, 9, 11, 3)), ((1, 6, 9, 4), (1, 8, 5, 4))])
def test_add_with_different_extents(ctx, s0, s1):
    """Output = max extent; the sum runs over the first input's extent, the second reads 0 outside its own (vk_add.comp:47-49)."""
    import shadernn_amd as snn

    a, b = _rand(s0, 21), _rand(s1, 22)
    H, W = max(s0[1], s1[1]), max(s0[2], s1[2])
    y, desc = _run(ctx, snn.add_plan(ctx, s0[0], H, W, s0[3], act="relu"), a, b)
    want = O.add_act(a, b, "relu")
    assert y.shape == want.shape == (s0[0], H, W, s0[3])
    np.testing.assert_allclose(y, want, err_msg=desc, **TOL)
    ref = np.zeros_like(want)
    ref[:, : s0[1], : s0[2]] = a
    ref[:, : min(s0[1], s1[1]), : min(s0[2], s1[2])] += b[:, : min(s0[1], s1[1]), : min(s0[2], s1[2])]
    ref[:, : s0[1], : s0[2]] = np.maximum(ref[:, : s0[1], : s0[2]], 0)
    ref[:, s0[1]:, :] = 0
    ref[:, :, s0[2]:] = 0
    np.testing.assert_allclose(want, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("mode", ["constant", "replicate", "reflect"])
@pytest.mark.parametrize("n,h,w,ic,oc,k,s,pads,conv_pad", [(1, 40, 48, 32, 32, 3, 1, (1, 1, 1, 1), "valid"), (2, 21, 27, 3, 32, 9, 1, (4, 4, 4, 4), "valid"),
                                                          (1, 33, 29, 32, 64, 3, 2, (1, 1, 1, 1), "valid"), (1, 18, 22, 64, 32, 3, 1, (2, 1, 3, 0), "same")])
def test_pad_conv_chain_fusion(ctx, n, h, w, ic, oc, k, s, pads, conv_pad, mode, dtype):
    """Chain rule D: Pad + Conv2D as ONE launch (the conv stages its tiles from the unpadded tensor), vs the oracle's pad then conv --
    including the reference's swapped x/y pad offsets and the unshrunk "valid" output (Q20)."""
    import shadernn_amd as snn

    dt = snn.F16 if dtype == "f16" else snn.F32
    x, wt, b = _rand((n, h, w, ic), 1), _rand((oc, ic, k, k), 2, 1.0 / np.sqrt(ic * k * k)), _rand((oc,), 3, 0.1)
    ph, pw = h + pads[0] + pads[1], w + pads[2] + pads[3]
    cp = O.padding_offsets(conv_pad, k)
    pad = snn.pad_plan(ctx, n, h, w, ic, pads, mode)
    conv = snn.conv2d_plan(ctx, n, ph, pw, wt, b, stride=s, pads=cp, act="relu", dtype=dt)
    chain = snn.chain_plan(ctx, [pad, conv])
    assert chain.num_steps() == 1 and "+pad(%s)" % mode in chain.describe(), chain.describe()
    xt = snn.Tensor.from_numpy(ctx, x, dtype=dt)
    y = chain(xt)
    if dtype == "f16":
        want = O._h(O.conv2d(O._h(O.pad(O._h(x), pads, mode)), O._h(wt), b, s, cp, "constant", "relu", 0.0, None))
        np.testing.assert_allclose(y.numpy(), want, err_msg=chain.describe(), rtol=2e-3, atol=2e-3)
    else:
        want = O.conv2d(O.pad(x, pads, mode), wt, b, s, cp, "constant", "relu", 0.0, None)
        np.testing.assert_allclose(y.numpy(), want, err_msg=chain.describe(), **TOL)
    two = conv(pad(xt))  # the unfused pair on the same input
    np.testing.assert_allclose(y.numpy(), two.numpy(), rtol=2e-3 if dtype == "f16" else 1e-5, atol=2e-3 if dtype == "f16" else 1e-5)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("n,h,w,ic,oc,k,s,act,add_act,split", [(2, 28, 28, 64, 64, 3, 1, "", "relu", None), (2, 14, 14, 96, 32, 1, 1, "", "", None),
                                                               (3, 7, 7, 512, 512, 3, 1, "", "relu", "2"), (1, 19, 23, 32, 40, 3, 1, "relu6", "leakyRelu", None),
                                                               (2, 16, 16, 64, 36, 1, 1, "", "relu", None)])
def test_conv_add_chain_fusion(ctx, monkeypatch, n, h, w, ic, oc, k, s, act, add_act, split, dtype):
    """Chain rule E: Conv2D + Add as one launch (residual added in the convolution's epilogue / the split-K reduce pass), vs conv then add."""
    import shadernn_amd as snn

    if split:
        monkeypatch.setenv("SNNHIP_CONV_SPLITK", split)
    dt = snn.F16 if dtype == "f16" else snn.F32
    x, wt, b = _rand((n, h, w, ic), 1), _rand((oc, ic, k, k), 2, 1.0 / np.sqrt(ic * k * k)), _rand((oc,), 3, 0.1)
    skip = _rand((n, h, w, oc), 4)
    bn = _bn(oc, 5)
    pads = O.padding_offsets("same", k)
    conv = snn.conv2d_plan(ctx, n, h, w, wt, b, stride=s, pads=pads, act=act, bn=bn, dtype=dt)
    add = snn.add_plan(ctx, n, h, w, oc, act=add_act, leaky=0.2)
    fused = snn.chain_plan(ctx, [conv, add])
    assert fused.num_steps() == 1 and "+add" in fused.describe(), fused.describe()
    if split:
        assert "splitK=2" in fused.describe()
    xt, st = snn.Tensor.from_numpy(ctx, x, dtype=dt), snn.Tensor.from_numpy(ctx, skip, dtype=dt)
    y = fused([xt, st]).numpy()
    two = add([conv(xt), st]).numpy()
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == "f16" else TOL
    if dtype == "f16":
        c = O._h(O.conv2d(O._h(x), O._h(wt), b, s, pads, "constant", act, 0.0, bn))
        want = O._h(O.add_act(c, O._h(skip), add_act, 0.2))
    else:
        want = O.add_act(O.conv2d(x, wt, b, s, pads, "constant", act, 0.0, bn), skip, add_act, 0.2)
    np.testing.assert_allclose(y, want, err_msg=fused.describe(), **tol)
    np.testing.assert_allclose(y, two, rtol=1e-6 if dtype == "f32" else 2e-3, atol=1e-6 if dtype == "f32" else 2e-3)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("with_pad", [True, False])
def test_upsample_pad_conv_chain_fusion(ctx, with_pad, dtype):
    """Rule D with a nearest x2 UpSampling2D in front: [UpSampling2D, Pad(reflect), Conv2D] (Candy's decoder) as one convolution launch."""
    import shadernn_amd as snn

    dt = snn.F16 if dtype == "f16" else snn.F32
    n, h, w, ic, oc = 1, 17, 23, 64, 32
    x, wt, b = _rand((n, h, w, ic), 1), _rand((oc, ic, 3, 3), 2, 1.0 / np.sqrt(ic * 9)), _rand((oc,), 3, 0.1)
    up = snn.upsample_plan(ctx, n, h, w, ic, 2.0, "nearest")
    plans, ph, pw = [up], 2 * h, 2 * w
    if with_pad:
        plans.append(snn.pad_plan(ctx, n, ph, pw, ic, (1, 1, 1, 1), "reflect"))
        ph, pw = ph + 2, pw + 2
    cp = O.padding_offsets("valid" if with_pad else "same", 3)
    plans.append(snn.conv2d_plan(ctx, n, ph, pw, wt, b, stride=1, pads=cp, act="relu", dtype=dt))
    chain = snn.chain_plan(ctx, plans)
    assert chain.num_steps() == 1 and "+upsample(x2)" in chain.describe(), chain.describe()
    xt = snn.Tensor.from_numpy(ctx, x, dtype=dt)
    y = chain(xt).numpy()
    q = O._h if dtype == "f16" else (lambda a: a)
    t = O.upsample(q(x), 2.0, "nearest")
    if with_pad:
        t = O.pad(t, (1, 1, 1, 1), "reflect")
    want = q(O.conv2d(t, q(wt), b, 1, cp, "constant", "relu", 0.0, None))
    np.testing.assert_allclose(y, want, err_msg=chain.describe(), **(dict(rtol=2e-3, atol=2e-3) if dtype == "f16" else TOL))


@pytest.mark.parametrize("with_pad", [False, True])
@pytest.mark.parametrize("n,h,w,ic,oc,k,s", [(1, 40, 56, 32, 64, 3, 1), (2, 37, 45, 64, 32, 3, 1), (1, 64, 64, 32, 64, 3, 2), (1, 33, 70, 16, 128, 3, 1)])
def test_conv_instancenorm_chain_fusion(ctx, monkeypatch, n, h, w, ic, oc, k, s, with_pad):
    """Chain rule F (fp16, opt-in): [Pad ->] Conv2D -> InstanceNorm as one step; the norm's statistics come from the convolution's per-tile
    records (fold + one in-place sweep) instead of a sweep over the tensor.  Checked against the unfused launches and the oracle."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_NORM_FUSION", "1")

    dt = snn.F16
    x, wt, b = _rand((n, h, w, ic), 1), _rand((oc, ic, k, k), 2, 1.0 / np.sqrt(ic * k * k)), _rand((oc,), 3, 0.5)
    beta, gamma = _rand((oc,), 4, 0.3), 1.0 + _rand((oc,), 5, 0.2)
    plans, hp, wp, cp = [], h, w, O.padding_offsets("same", k)
    if with_pad:
        pads = (1, 1, 1, 1)
        plans.append(snn.pad_plan(ctx, n, h, w, ic, pads, "reflect"))
        hp, wp, cp = h + 2, w + 2, (0, 0, 0, 0)
    conv = snn.conv2d_plan(ctx, n, hp, wp, wt, b, stride=s, pads=cp, act="", dtype=dt)
    oh, ow = conv.out_shape()[1:3]
    norm = snn.instancenorm_plan(ctx, n, oh, ow, oc, beta, gamma, act="relu")
    plans += [conv, norm]
    fused = snn.chain_plan(ctx, plans)
    assert fused.num_steps() == 1 and "tile stats" in fused.describe(), fused.describe()
    xt = snn.Tensor.from_numpy(ctx, x, dtype=dt)
    y = fused(xt).numpy()
    t = xt
    for pl in plans:
        t = pl(t)
    two = t.numpy()
    xin = O._h(x)
    if with_pad:
        xin = O.pad(xin, (1, 1, 1, 1), "reflect")
    c = O._h(O.conv2d(xin, O._h(wt), b, s, cp, "constant", "", 0.0, None))
    want = O._h(O.instancenorm(c, beta, gamma, "relu"))
    np.testing.assert_allclose(y, want, err_msg=fused.describe(), rtol=4e-3, atol=4e-3)
    np.testing.assert_allclose(y, two, rtol=3e-3, atol=3e-3)


def test_conv_instancenorm_not_fused_in_fp32(ctx, monkeypatch):
    """fp32 convolutions have no LDS output tile to take statistics from: the pair stays two launches (and correct)."""
    import shadernn_amd as snn

    monkeypatch.setenv("SNNHIP_NORM_FUSION", "1")

    n, h, w, ic, oc = 1, 24, 24, 16, 32
    x, wt, b = _rand((n, h, w, ic), 1), _rand((oc, ic, 3, 3), 2, 0.1), _rand((oc,), 3, 0.1)
    beta, gamma = _rand((oc,), 4, 0.3), 1.0 + _rand((oc,), 5, 0.2)
    pad = snn.pad_plan(ctx, n, h, w, ic, (1, 1, 1, 1), "reflect")
    conv = snn.conv2d_plan(ctx, n, h + 2, w + 2, wt, b, stride=1, pads=(0, 0, 0, 0), act="")
    oh, ow = conv.out_shape()[1:3]
    norm = snn.instancenorm_plan(ctx, n, oh, ow, oc, beta, gamma, act="")
    chain = snn.chain_plan(ctx, [pad, conv, norm])
    assert chain.num_steps() == 2, chain.describe()
    y = chain(snn.Tensor.from_numpy(ctx, x)).numpy()
    want = O.instancenorm(O.conv2d(O.pad(x, (1, 1, 1, 1), "reflect"), wt, b, 1, (0, 0, 0, 0), "constant", "", 0.0, None), beta, gamma, "")
    np.testing.assert_allclose(y, want, **TOL)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("n,h,w,c,act,add_act", [(1, 24, 40, 32, "", ""), (2, 19, 23, 64, "relu", ""), (1, 45, 80, 128, "", "relu"), (3, 7, 9, 4, "", "")])
def test_instancenorm_add_graph_fusion(ctx, monkeypatch, n, h, w, c, act, add_act, dtype):
    """Graph rule H (the residual blocks of the style networks): InstanceNorm -> Add(., skip) becomes the norm's own launches with the addition in
    the normalise sweep.  Same rounding points as the two launches: bit-identical to the unfused pair, and within tolerance of the oracle."""
    import shadernn_amd as snn

    dt = snn.F16 if dtype == "f16" else snn.F32
    x, skip = 2.0 * _rand((n, h, w, c), 1) + 0.3, _rand((n, h, w, c), 2)
    beta, gamma = _rand((c,), 4, 0.3), 1.0 + _rand((c,), 5, 0.2)
    pre = snn.activation_plan(ctx, n, h, w, c, "tanh")          # a producer in front, so that the norm's input is a graph tensor
    norm = snn.instancenorm_plan(ctx, n, h, w, c, beta, gamma, act=act)
    add = snn.add_plan(ctx, n, h, w, c, act=add_act)
    for order in ([1, -2], [-2, 1]):                              # the skip connection may be either summand
        nodes = [(pre, [-1], False), (norm, [0], False), (add, order, True)]
        fused = snn.graph_fuse(ctx, nodes)
        assert fused[1][0] is None and "+add" in fused[2][0].describe() and fused[2][1] == [0, -2], (fused[2][0].describe(), fused[2][1])
        xt, st = snn.Tensor.from_numpy(ctx, x, dtype=dt), snn.Tensor.from_numpy(ctx, skip, dtype=dt)
        t0 = pre(xt)
        y = fused[2][0]([t0, st]).numpy()
        two = add([norm(t0), st]).numpy()
        np.testing.assert_array_equal(y, two)
    q = O._h if dtype == "f16" else (lambda a: a)
    t = q(O.add_act(q(x), None, "tanh"))
    want = q(O.add_act(q(O.instancenorm(t, beta, gamma, act)), q(skip), add_act))
    np.testing.assert_allclose(y, want, **(dict(rtol=4e-3, atol=4e-3) if dtype == "f16" else TOL))
    # a skip smaller than the norm (Candy's residual blocks under the reference's size rule): the ragged-Add rule, both input orders
    if h > 8:
        small = np.ascontiguousarray(skip[:, : h - 4, : w - 4, :])
        st = snn.Tensor.from_numpy(ctx, small, dtype=dt)
        for order in ([1, -2], [-2, 1]):
            fused = snn.graph_fuse(ctx, [(pre, [-1], False), (norm, [0], False), (add, order, True)])
            assert fused[1][0] is None and ("residual first" in fused[2][0].describe()) == (order[0] == -2)
            y = fused[2][0]([t0, st]).numpy()
            two = add([norm(t0), st] if order[0] == 1 else [st, norm(t0)]).numpy()
            np.testing.assert_array_equal(y, two)
            nrm = q(O.instancenorm(t, beta, gamma, act))
            want = O.add_act(nrm, None, add_act) if order[0] == 1 else np.zeros_like(nrm)
            want[:, : h - 4, : w - 4, :] = O.add_act(nrm[:, : h - 4, : w - 4, :], q(small), add_act)
            np.testing.assert_allclose(y, q(want), **(dict(rtol=4e-3, atol=4e-3) if dtype == "f16" else TOL))
    # a norm that somebody else reads as well stays a launch of its own
    nodes = [(pre, [-1], False), (norm, [0], True), (add, [1, -2], True)]
    assert all(p is not None for p, _ in snn.graph_fuse(ctx, nodes))
    monkeypatch.setenv("SNNHIP_NO_ADD_FUSION", "1")
    nodes = [(pre, [-1], False), (norm, [0], False), (add, [1, -2], True)]
    assert all(p is not None for p, _ in snn.graph_fuse(ctx, nodes))


@pytest.mark.parametrize("with_pad,with_up", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("n,h,w,ic,oc,k,s,act", [(1, 40, 56, 32, 64, 3, 2, "relu"), (2, 37, 45, 64, 128, 3, 2, "relu"), (1, 33, 70, 16, 32, 3, 1, ""), (2, 19, 23, 64, 64, 3, 1, "leakyRelu"),
                                                (1, 30, 30, 8, 16, 5, 1, "relu"), (2, 21, 70, 32, 3, 9, 1, "relu"), (1, 17, 40, 16, 3, 7, 1, "")])
def test_instancenorm_conv_chain_fusion(ctx, monkeypatch, n, h, w, ic, oc, k, s, act, with_pad, with_up):
    """Graph rule I (fp16): InstanceNorm -> [UpSampling] -> [Pad] -> Conv2D as the norm's statistics sweep + fold and ONE convolution launch that
    normalises while it stages its input.  Same arithmetic and rounding points as the separate launches: bit-identical to them (and within
    tolerance of the oracle); zero padding of a 'same' convolution stays zero."""
    import shadernn_amd as snn

    if with_up and (s != 1 or oc == 3):
        pytest.skip("the upsampling staging path: stride-1 layers of the general MFMA kernel")
    monkeypatch.setenv("SNNHIP_CONV_WIDE", "0")  # (the wide kernel gets its own cases in test_conv_wide_gpu.py)
    dt = snn.F16
    x = 1.5 * _rand((n, h, w, ic), 1) + 0.2
    wt, b = _rand((oc, ic, k, k), 2, 1.0 / np.sqrt(ic * k * k)), _rand((oc,), 3, 0.5)
    beta, gamma = _rand((ic,), 4, 0.3), 1.0 + _rand((ic,), 5, 0.2)
    norm = snn.instancenorm_plan(ctx, n, h, w, ic, beta, gamma, act=act, leaky=0.1)
    plans, hh, ww = [norm], h, w
    if with_up:
        plans.append(snn.upsample_plan(ctx, n, h, w, ic, 2.0, "nearest"))
        hh, ww = 2 * h, 2 * w
    cp = O.padding_offsets("same", k)
    if with_pad:
        pd = k // 2
        plans.append(snn.pad_plan(ctx, n, hh, ww, ic, (pd, pd, pd, pd), "reflect"))
        hh, ww, cp = hh + 2 * pd, ww + 2 * pd, (0, 0, 0, 0)
    plans.append(snn.conv2d_plan(ctx, n, hh, ww, wt, b, stride=s, pads=cp, act="relu", dtype=dt))
    fused = snn.chain_plan(ctx, plans)
    assert fused.num_steps() == 1 and "instancenorm(statistics sweep + fold) -> instancenorm(act=" in fused.describe(), fused.describe()
    xt = snn.Tensor.from_numpy(ctx, x, dtype=dt)
    y = fused(xt).numpy()
    t = xt
    for pl in plans:
        t = pl(t)
    np.testing.assert_array_equal(y, t.numpy())
    ref = O._h(O.instancenorm(O._h(x), beta, gamma, act, 0.1))
    if with_up:
        ref = O.upsample(ref, 2.0, "nearest")
    if with_pad:
        ref = O.pad(ref, (k // 2,) * 4, "reflect")
    want = O._h(O.conv2d(ref, O._h(wt), b, s, cp, "constant", "relu", 0.0, None))
    np.testing.assert_allclose(y, want, err_msg=fused.describe(), rtol=6e-3, atol=6e-3)
    monkeypatch.setenv("SNNHIP_NO_NORM_FOLD", "1")
    assert "statistics sweep" not in snn.chain_plan(ctx, plans).describe() if (with_pad or with_up) else True
