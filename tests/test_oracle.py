"""CPU tests of the oracle itself: the two independent walks agree, and both agree with torch's CPU conv2d (the same
math as the ncnn naive layers the reference's own unit tests compare against, demo/test/unittest/convolutionTest.cpp:44-94)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle_lib as O


def _rand(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


def _bn(c, seed):
    r = np.random.default_rng(seed)
    return {"beta": r.uniform(-0.1, 0.1, c).astype(np.float32), "gamma": r.uniform(0.5, 1.5, c).astype(np.float32),
            "mean": r.uniform(-0.1, 0.1, c).astype(np.float32), "var": r.uniform(0.5, 1.5, c).astype(np.float32)}


def torch_conv(x, w, b, stride, pads, pad_mode, groups=1):
    """pads = (T,B,L,R) in the reference's order; its kernels offset x by T and y by L (quirk Q3)."""
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    T, B, L, R = pads
    k = w.shape[-1]
    # reference: x offset = padT, y offset = padL; output size from T+B on both axes
    H, W = x.shape[1], x.shape[2]
    OH, OW = O.out_dim(H, k, stride, T, B), O.out_dim(W, k, stride, T, B)
    needH = (OH - 1) * stride + k
    needW = (OW - 1) * stride + k
    padl, padt = T, L
    padr, padb = max(needW - W - padl, 0), max(needH - H - padt, 0)
    mode = {"constant": "constant", "replicate": "replicate", "reflect": "reflect", "none": "constant"}[pad_mode]
    xp = F.pad(xt, (padl, padr, padt, padb), mode=mode)
    y = F.conv2d(xp, torch.from_numpy(w), None if b is None else torch.from_numpy(b), stride=stride, groups=groups)
    y = y[:, :, :OH, :OW]
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def test_padding_rule_Q1():
    assert O.padding_offsets("same", 3) == (1, 1, 1, 1)
    assert O.padding_offsets("same", 5) == (2, 2, 2, 2)
    assert O.padding_offsets("same", 1) == (0, 0, 0, 0)
    assert O.padding_offsets("same", 4) == (1, 2, 1, 2)  # even kernel: T-=1, L-=1 (conv2d.cpp:62-65)
    assert O.padding_offsets("same", 2) == (0, 1, 0, 1)
    assert O.padding_offsets("valid", 3) == (0, 0, 0, 0)
    assert O.padding_offsets("none", 7) == (0, 0, 0, 0)
    assert O.padding_offsets("3", 7) == (3, 3, 3, 3)


def test_out_dim_rule_Q2():
    assert O.out_dim(224, 3, 1, 1, 1) == 224
    assert O.out_dim(1080, 5, 1, 2, 2) == 1080
    assert O.out_dim(224, 7, 2, 3, 3) == 112
    assert O.out_dim(56, 1, 2, 0, 0) == 28
    assert O.out_dim(7, 1, 2, 0, 0) == 4       # 3.5 + 0.5
    assert O.out_dim(9, 1, 2, 0, 0) == 5
    # 'valid': translation = 1 + (0-3)/1 = -2 is clamped by max(0, .) in genericlayer.cpp:77-78 => the reference
    # reports an UNSHRUNK 8 (the shader then reads zeros past the edge). Restated faithfully (quirk Q20).
    assert O.out_dim(8, 3, 1, 0, 0) == 8
    assert O.out_dim(9, 3, 2, 0, 0) == 4
    assert O.out_dim(112, 3, 2, 1, 1) == 56
    assert O.out_dim(8, 4, 1, 1, 2) == 8       # even kernel uses T+B-1


@pytest.mark.parametrize("ic,oc,k,stride,act", [(3, 4, 3, 1, "relu"), (4, 5, 3, 2, ""), (5, 1, 3, 1, "tanh"), (16, 64, 3, 1, "leakyRelu"),
                                                (128, 1, 1, 1, ""), (8, 12, 1, 2, "relu6"), (1, 16, 5, 1, "sigmoid"), (6, 7, 7, 2, "SiLU")])
@pytest.mark.parametrize("pad_mode", ["constant", "replicate", "reflect"])
def test_conv_walks_agree_and_match_torch(ic, oc, k, stride, act, pad_mode):
    x = _rand((1, 9, 11, ic), 1)
    w = (_rand((oc, ic, k, k), 2) / np.sqrt(ic * k * k)).astype(np.float32)
    b = (_rand((oc,), 3) * 0.1).astype(np.float32)
    pads = O.padding_offsets("same", k)
    y1 = O.conv2d(x, w, b, stride, pads, pad_mode, act, 0.1)
    y2 = O.conv2d_texel(x, w, b, stride, pads, pad_mode, act, 0.1)
    assert y1.shape == y2.shape
    np.testing.assert_allclose(y1, y2, rtol=2e-5, atol=2e-6)
    # independent statement (no activation / BN): torch
    y0 = O.conv2d(x, w, b, stride, pads, pad_mode, "", 0.0)
    pm = "constant" if k == 1 else pad_mode  # the 1x1 shader has no padding at all
    yt = torch_conv(x, w, b, stride, pads, pm)
    np.testing.assert_allclose(y0, yt, rtol=2e-5, atol=2e-5)


def test_conv_bn_identity_matches_reference_test_setup():
    # convolutionTest.cpp:101-125 uses gamma=1, mean=0, var=1, beta=0; the shader's eps=1e-3 still scales by 1/sqrt(1.001)
    x = _rand((1, 8, 8, 4), 5)
    w = _rand((4, 4, 3, 3), 6)
    bn = {"beta": np.zeros(4, np.float32), "gamma": np.ones(4, np.float32), "mean": np.zeros(4, np.float32), "var": np.ones(4, np.float32)}
    y = O.conv2d(x, w, None, 1, (1, 1, 1, 1), "constant", "", 0.0, bn)
    y0 = O.conv2d(x, w, None, 1, (1, 1, 1, 1), "constant", "", 0.0, None)
    np.testing.assert_allclose(y, y0 / np.sqrt(np.float32(1.001)), rtol=1e-6, atol=1e-6)
    yt = O.conv2d_texel(x, w, None, 1, (1, 1, 1, 1), "constant", "", 0.0, bn)
    np.testing.assert_allclose(y, yt, rtol=2e-5, atol=2e-6)


def test_silu_quirk_differs_only_off_group_heads():
    x = _rand((1, 4, 8, 4), 7)
    w = _rand((4, 4, 3, 3), 8)
    y_ok = O.conv2d(x, w, None, act="SiLU")
    y_q = O.conv2d(x, w, None, act="SiLU_quirk")
    y_qt = O.conv2d_texel(x, w, None, act="SiLU_quirk")
    np.testing.assert_allclose(y_q, y_qt, rtol=2e-5, atol=2e-6)
    np.testing.assert_array_equal(y_ok[:, :, 0::4, :], y_q[:, :, 0::4, :])
    assert np.abs(y_ok[:, :, 1::4, :] - y_q[:, :, 1::4, :]).max() > 1e-3


@pytest.mark.parametrize("c,k,stride", [(8, 1, 2), (8, 3, 1), (32, 3, 2), (96, 3, 1), (5, 3, 1), (7, 5, 2)])
def test_depthwise_walks_agree_and_match_torch(c, k, stride):
    x = _rand((1, 9, 9, c), 11)
    w = _rand((c, k, k), 12)
    b = _rand((c,), 13)
    bn = _bn(c, 14)
    pads = O.padding_offsets("same", k)
    y1 = O.depthwise(x, w, b, stride, pads, "relu6", 0.0, bn)
    y2 = O.depthwise_texel(x, w, b, stride, pads, "relu6", 0.0, bn)
    np.testing.assert_allclose(y1, y2, rtol=2e-5, atol=2e-6)
    y0 = O.depthwise(x, w, b, stride, pads)
    yt = torch_conv(x, w[:, None], b, stride, pads, "constant", groups=c)
    np.testing.assert_allclose(y0, yt, rtol=2e-5, atol=2e-5)


def test_depthwise_reference_unit_test_case():
    # depthwiseConv2DTest.cpp:316: w=9,h=9,c=8,kernel=1,stride=2,pad=0,bias, input all 1.0
    x = np.ones((1, 9, 9, 8), np.float32)
    w = _rand((8, 1, 1), 21)
    b = _rand((8,), 22)
    y = O.depthwise(x, w, b, 2, (0, 0, 0, 0))
    assert y.shape == (1, 5, 5, 8)
    np.testing.assert_allclose(y, np.broadcast_to(w[:, 0, 0] + b, y.shape), rtol=1e-6)


def test_subpixel_matches_pixel_shuffle():
    x = _rand((2, 6, 5, 4), 31)
    y = O.subpixel(x, 2, 0)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    yt = torch.tanh(F.pixel_shuffle(xt, 2)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y, yt, rtol=1e-6, atol=1e-6)
    # Vulkan quirk: 1 texel deep => always channel 0
    yq = O.subpixel(x, 2, 1)
    np.testing.assert_allclose(yq[:, ::2, ::2, 0], np.tanh(x[..., 0]), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(yq[:, 1::2, 1::2, 0], np.tanh(x[..., 0]), rtol=1e-6, atol=1e-6)


def test_medium_precision_is_truncation():
    for v in [0.1, -0.1, 1.0, 65504.0, 1e-3, 3.14159, -2.71828, 1e-9, 1e9]:
        got = O.to_medium_precision(v)
        bits = np.float32(v).view(np.uint32)
        e = int((bits >> 23) & 0xFF) - 127 + 15
        if e <= 0:
            want = 0.0 * np.sign(v)
        elif e >= 31:
            want = np.sign(v) * np.inf
        else:
            want = np.uint32(bits & np.uint32(0xFFFFE000)).view(np.float32)
        assert got == want or (np.isinf(got) and np.isinf(want)), (v, got, want)


def test_dense_semantics_Q8_and_unknown_activation_Q19():
    w = np.array([1, -2, 3, 4, 5, 6], np.float32)
    b = np.array([0.5, -100], np.float32)
    x = np.array([[1, 2, 3]], np.float32)
    np.testing.assert_allclose(O.dense(x, w, 2, b, "relu"), [[6.5, 0.0]])
    np.testing.assert_allclose(O.dense(x, w, 2, b, ""), [[6.5, -68.0]])
    np.testing.assert_allclose(O.dense(x, w, 2, b, "linear"), [[6.5, 0.0]])  # unordered_map::operator[] -> RELU


@pytest.mark.parametrize("act", ["relu", "", "sigmoid", "tanh", "softmax", "leakyRelu", "SiLU", "linear"])
def test_dense_against_reference_eigen_path(act):
    """PINS the dense oracle: oracle/_ref/ref_dense is the reference's own CPUCommonUtil<float> (Eigen)."""
    rng = np.random.default_rng(41)
    In, Out = 11, 5  # denseTest.cpp:111
    w = rng.standard_normal(In * Out).astype(np.float32)
    b = rng.standard_normal(Out).astype(np.float32)
    x = rng.standard_normal(In).astype(np.float32)
    ref = O.ref_dense(In, Out, act, 0.3, w, b, x)
    if ref is None:
        pytest.skip("oracle/_ref/ref_dense not built (needs /root/reference)")
    got = O.dense(x[None], w, Out, b, act, 0.3)[0]
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)


# ---- element-wise / pooling / shape operators: the C restatements vs torch CPU ops ------------------------------

def test_pool_restatement_against_torch():
    import torch
    import torch.nn.functional as F

    x = np.random.default_rng(3).standard_normal((2, 12, 14, 6)).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    # k=2 s=2 valid: identical to torch; k=3 s=2 "same": the reference pads bottom/right only (top/left forced to 0)
    np.testing.assert_allclose(O.pool2d(x, 2, 2, "max", same=False), F.max_pool2d(xt, 2, 2).permute(0, 2, 3, 1).numpy(), rtol=1e-6)
    np.testing.assert_allclose(O.pool2d(x, 2, 2, "avg", same=False), F.avg_pool2d(xt, 2, 2).permute(0, 2, 3, 1).numpy(), rtol=1e-6, atol=1e-7)
    want = F.max_pool2d(F.pad(xt, (0, 1, 0, 1), value=-1e30), 3, 2).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(O.pool2d(x, 3, 2, "max", same=True), want, rtol=1e-6)
    assert O.pool_out_dim(112, 3, 2, True) == 56 and O.pool_out_dim(12, 3, 2, False) == 6  # Q20: "valid" is not shrunk
    np.testing.assert_allclose(O.global_avgpool(x)[:, 0, 0], x.mean(axis=(1, 2)), rtol=1e-5, atol=1e-6)


def test_pad_upsample_instancenorm_restatements_against_torch():
    import torch
    import torch.nn.functional as F

    x = np.random.default_rng(4).standard_normal((2, 9, 11, 5)).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    for mode, tmode in (("constant", "constant"), ("replicate", "replicate"), ("reflect", "reflect")):
        want = F.pad(xt, (3, 3, 3, 3), mode=tmode).permute(0, 2, 3, 1).numpy()
        np.testing.assert_array_equal(O.pad(x, (3, 3, 3, 3), mode), want)
    np.testing.assert_array_equal(O.upsample(x, 2.0, "nearest"), F.interpolate(xt, scale_factor=2, mode="nearest").permute(0, 2, 3, 1).numpy())
    want = F.interpolate(xt, scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(O.upsample(x, 2.0, "bilinear"), want, rtol=1e-5, atol=1e-6)
    g, b = np.linspace(0.5, 1.5, 5, dtype=np.float32), np.linspace(-0.2, 0.2, 5, dtype=np.float32)
    want = F.instance_norm(xt, weight=torch.from_numpy(g), bias=torch.from_numpy(b), eps=1e-5).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(O.instancenorm(x, b, g), want, rtol=1e-4, atol=1e-5)
    # asymmetric pads expose the reference's swapped offsets: x is shifted by the TOP pad, y by the LEFT pad
    y = O.pad(x, (1, 0, 2, 0), "constant")
    assert y.shape == (2, 10, 13, 5)
    np.testing.assert_array_equal(y[:, 2:, 1:12, :], x[:, :8, :, :])


def test_add_batchnorm_restatements():
    a = np.random.default_rng(5).standard_normal((1, 3, 4, 6)).astype(np.float32)
    b = np.random.default_rng(6).standard_normal((1, 3, 4, 6)).astype(np.float32)
    np.testing.assert_allclose(O.add_act(a, b, "relu"), np.maximum(a + b, 0), rtol=1e-7)
    bn = {"beta": np.full(6, 0.1, np.float32), "gamma": np.full(6, 2.0, np.float32), "mean": np.full(6, 0.5, np.float32), "var": np.full(6, 4.0, np.float32)}
    np.testing.assert_allclose(O.batchnorm(a, bn), 2.0 / np.sqrt(np.float32(4.001)) * (a - 0.5) + 0.1, rtol=1e-6, atol=1e-6)
