"""Concatenate / Unary / Conv2DTranspose / Calculate, input resize + normalise, 8-bit image -> tensor, argmax, YOLO decode.
CPU part: the oracle restatements against the reference shader restated line by line (transposed conv) and against torch; GPU part: the HIP
kernels through the C-ABI against the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from test_ops_gpu import TOL, _bn, _rand


# ------------------------------------------------------------------------------------------------ oracle (CPU)
def test_deconv_formula_matches_the_reference_shader_restated(built):
    """snn_oracle_deconv2d (general k, s) vs the k=4 s=2 compute shader restated line by line, incl. a ragged channel count."""
    for ic, oc, h, w in [(4, 4, 5, 6), (7, 5, 3, 4), (9, 2, 4, 3)]:
        x, wt, b = _rand((1, h, w, ic), 11), _rand((oc, ic, 4, 4), 12, 0.3), _rand((oc,), 13, 0.1)
        got = O.deconv2d(x, wt, b, stride=2, same=True)
        want = O.deconv4x4s2_shader(x[0], wt, b)
        np.testing.assert_allclose(got[0], want, rtol=1e-5, atol=1e-5)


def test_deconv_is_torch_conv_transpose_with_flipped_kernel(built):
    import torch
    import torch.nn.functional as F

    for k, s, same in [(4, 2, True), (3, 1, True), (4, 2, False), (2, 2, True), (5, 1, False)]:
        x, wt, b = _rand((2, 5, 7, 3), 21), _rand((4, 3, k, k), 22, 0.3), _rand((4,), 23, 0.1)
        got = O.deconv2d(x, wt, b, stride=s, same=same, act="relu")
        p = (k - s) // 2 if same else 0
        # correlation with the zero-stuffed input == conv_transpose2d with the spatially flipped kernel, padding k-1-(k-1-p) = p
        tw = torch.from_numpy(wt).flip(2, 3).permute(1, 0, 2, 3).contiguous()
        ref = F.conv_transpose2d(torch.from_numpy(x).permute(0, 3, 1, 2), tw, torch.from_numpy(b), stride=s, padding=p)
        ref = torch.relu(ref).permute(0, 2, 3, 1).numpy()
        oh = got.shape[1]
        np.testing.assert_allclose(got, ref[:, :oh, :got.shape[2], :], rtol=1e-4, atol=1e-5)


def test_concat_is_by_texel_plane(built):
    a, b = _rand((1, 2, 3, 8), 1), _rand((1, 2, 3, 4), 2)
    np.testing.assert_array_equal(O.concat(a, b), np.concatenate([a, b], axis=-1))
    a6, b6 = _rand((1, 2, 3, 6), 3), _rand((1, 2, 3, 6), 4)
    y = O.concat(a6, b6)  # 12 output channels = 3 planes: in0 fills planes 0-1 (6 real + 2 zero lanes), in1's first plane lands in plane 2
    np.testing.assert_array_equal(y[..., :6], a6)
    np.testing.assert_array_equal(y[..., 6:8], 0)
    np.testing.assert_array_equal(y[..., 8:12], b6[..., :4])


def test_unary_calculate_argmax_image(built):
    x = _rand((1, 3, 4, 5), 5, 2.0)
    np.testing.assert_array_equal(O.unary(x, "copy"), x)
    np.testing.assert_array_equal(O.unary(x, "fixed", 0.25), np.full_like(x, 0.25))
    np.testing.assert_array_equal(O.unary(x, "neg"), -x)
    np.testing.assert_allclose(O.unary(x, "rcp"), 1.0 / x, rtol=1e-6)
    np.testing.assert_array_equal(O.unary(x, "square"), x * x)
    np.testing.assert_allclose(O.unary(x, "exp"), np.exp(x), rtol=1e-6)
    np.testing.assert_array_equal(O.unary(x, "abs"), np.abs(x))
    c = _rand((1, 2, 2, 12), 6) + 3.0
    y = O.calculate(c, 4)
    np.testing.assert_allclose(y[..., :3], c[..., :3] / c[..., 8:9], rtol=1e-6)
    np.testing.assert_array_equal(y[..., 3], 0)
    v = np.array([0.1, 0.7, 0.7, -1.0], np.float32)
    assert O.argmax(v) == 1  # first of the equal maxima
    img = np.random.default_rng(7).integers(0, 256, (1, 3, 5, 3), dtype=np.uint8)
    t = O.image_u8(img, (10, 20, 30, 0), (0.5, 0.25, 2.0, 1.0))
    np.testing.assert_allclose(t[..., :3], (img.astype(np.float32) - np.float32([10, 20, 30])) * np.float32([0.5, 0.25, 2.0]))
    np.testing.assert_array_equal(t[..., 3], 1.0)
    g = O.image_u8(img[..., :1], (10, 0, 0, 0), (0.5, 1, 1, 1))
    np.testing.assert_array_equal(g[..., 1], np.float32(-5.0))


def test_resize_matches_torch_interpolate(built):
    import torch
    import torch.nn.functional as F

    x = _rand((2, 9, 7, 4), 8)
    for oh, ow in [(18, 14), (5, 4), (9, 7), (13, 20)]:
        got = O.resize(x, oh, ow, linear=True)
        ref = F.interpolate(torch.from_numpy(x).permute(0, 3, 1, 2), size=(oh, ow), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)
    got = O.resize(x, 18, 14, linear=False)
    np.testing.assert_array_equal(got, np.repeat(np.repeat(x, 2, axis=1), 2, axis=2))
    got = O.resize(x, 9, 7, means=(1, 2, 3, 4), norms=(2, 2, 0.5, 1))
    np.testing.assert_allclose(got, (x - np.float32([1, 2, 3, 4])) * np.float32([2, 2, 0.5, 1]), rtol=1e-6, atol=1e-6)


def test_yolo_decode_restatement():
    heads = [np.full((1, 13, 13, 18), -8.0, np.float32), np.full((1, 26, 26, 18), -8.0, np.float32)]
    heads[0][0, 6, 6, 6:12] = [0.0, 0.0, 0.0, 0.0, 6.0, 6.0]    # anchor mask 4: 135 x 169
    heads[0][0, 6, 6, 12:18] = [0.1, 0.0, 0.0, 0.0, 5.0, 5.0]   # overlapping box on anchor 5: suppressed or kept by IoU
    heads[1][0, 3, 20, 0:6] = [0.0, 0.0, 0.0, 0.0, 4.0, 9.0]    # anchor mask 1: 23 x 27
    boxes = O.yolo_decode(heads)
    assert len(boxes) == 3 and boxes[0][1] >= boxes[1][1] >= boxes[2][1]
    b = [bb for bb in boxes if abs(bb[4] - 135.0 / 416) < 1e-6][0]
    assert abs(b[2] + b[4] / 2 - 6.5 / 13) < 1e-6 and abs(b[5] - 169.0 / 416) < 1e-6
    assert abs(b[1] - 1.0 / (1.0 + np.exp(-6.0) * (1.0 + np.exp(-6.0)))) < 1e-6  # the reference's probability expression as written


# ------------------------------------------------------------------------------------------------ HIP through the C-ABI
def _run(ctx, plan, *xs, dtype=None):
    import shadernn_amd as snn

    dtype = snn.F32 if dtype is None else dtype
    ts = [snn.Tensor.from_numpy(ctx, x, dtype=dtype) for x in xs]
    od = plan.out_shape()
    yt = snn.Tensor(ctx, *od, dtype=dtype)
    plan.run(ts if len(ts) > 1 else ts[0], yt)
    y = yt.numpy()
    for t in ts + [yt]:
        t.free()
    desc = plan.describe()
    plan.destroy()
    return y, desc


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,c0,c1,oc", [(2, 9, 11, 16, 8, None), (1, 5, 7, 6, 6, None), (1, 4, 4, 3, 5, None), (1, 6, 5, 8, 8, 12), (1, 26, 26, 128, 256, None)])
def test_concat_matches_oracle(ctx, n, h, w, c0, c1, oc):
    import shadernn_amd as snn

    a, b = _rand((n, h, w, c0), 1), _rand((n, h, w, c1), 2)
    y, desc = _run(ctx, snn.concat_plan(ctx, n, h, w, c0, c1, oc), a, b)
    np.testing.assert_array_equal(y, O.concat(a, b, oc), err_msg=desc)
    yh, _ = _run(ctx, snn.concat_plan(ctx, n, h, w, c0, c1, oc), a, b, dtype=snn.F16)
    np.testing.assert_array_equal(yh, O.concat(O._h(a), O._h(b), oc))


@pytest.mark.gpu
@pytest.mark.parametrize("op", list(O.UNARY_OPS))
@pytest.mark.parametrize("shape", [(2, 6, 7, 12), (1, 5, 3, 7)])
def test_unary_matches_oracle(ctx, op, shape):
    import shadernn_amd as snn

    x = _rand(shape, 3, 2.0)
    y, desc = _run(ctx, snn.unary_plan(ctx, *shape, op, 0.75), x)
    np.testing.assert_allclose(y, O.unary(x, op, 0.75), err_msg=desc, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_calculate_matches_oracle(ctx):
    import shadernn_amd as snn

    x = _rand((2, 7, 9, 12), 4) + 3.0
    y, desc = _run(ctx, snn.calculate_plan(ctx, 2, 7, 9, 12, 4), x)
    np.testing.assert_allclose(y, O.calculate(x, 4), err_msg=desc, rtol=1e-6)
    with pytest.raises(snn.SnnHipError):
        snn.calculate_plan(ctx, 1, 4, 4, 8, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("linear", [True, False])
@pytest.mark.parametrize("n,h,w,c,oh,ow", [(1, 1080, 1920, 4, 416, 416), (2, 9, 7, 4, 18, 14), (1, 9, 7, 3, 5, 4), (1, 32, 32, 8, 32, 32)])
def test_resize_matches_oracle(ctx, linear, n, h, w, c, oh, ow):
    import shadernn_amd as snn

    x = _rand((n, h, w, c), 5)
    means, norms = (0.1, 0.2, 0.3, 0.0), (2.0, 0.5, 1.5, 1.0)
    y, desc = _run(ctx, snn.resize_plan(ctx, n, h, w, c, oh, ow, means, norms, linear), x)
    np.testing.assert_allclose(y, O.resize(x, oh, ow, means, norms, linear), err_msg=desc, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("sc", [1, 3, 4])
def test_image_u8_and_argmax(ctx, sc):
    import shadernn_amd as snn

    img = np.random.default_rng(sc).integers(0, 256, (2, 37, 53, sc), dtype=np.uint8)
    means, norms = (127.5, 120.0, 110.0, 0.0), (1 / 127.5, 1 / 64.0, 1 / 32.0, 1 / 255.0)
    plan = snn.image_u8_plan(ctx, 2, 37, 53, sc, means, norms)
    src = snn.Tensor(ctx, 2, 37, 53, sc, dtype=snn.U8)
    src.upload_u8(img)
    for dt in (snn.F32, snn.F16):
        dst = snn.Tensor(ctx, 2, 37, 53, 4, dtype=dt)
        plan.run(src, dst)
        want = O.image_u8(img, means, norms)
        np.testing.assert_allclose(dst.numpy(), O._h(want) if dt == snn.F16 else want, rtol=1e-6, atol=1e-6)
        flat = dst.numpy().reshape(2, -1)
        assert dst.argmax(0) == O.argmax(flat[0]) and dst.argmax(1) == O.argmax(flat[1])
        dst.free()
    with pytest.raises(snn.SnnHipError):  # an 8-bit tensor is only accepted by the image plan
        snn.activation_plan(ctx, 2, 37, 53, sc, "relu")(src)
    with pytest.raises(snn.SnnHipError):
        src.numpy()
    src.free()
    t = snn.Tensor.from_numpy(ctx, np.float32([0.1, 0.7, 0.7, -1.0]).reshape(1, 1, 1, 4))
    assert t.argmax() == 1
    t.free()


@pytest.mark.gpu
@pytest.mark.parametrize("k,s,same,ic,oc,act,bn", [(4, 2, True, 16, 16, "relu", False), (4, 2, True, 7, 5, "tanh", True), (3, 1, True, 8, 12, "", False),
                                                   (4, 2, False, 4, 4, "sigmoid", False), (2, 2, True, 32, 16, "leakyRelu", True), (5, 1, False, 3, 2, "", False)])
def test_deconv2d_matches_oracle(ctx, k, s, same, ic, oc, act, bn):
    import shadernn_amd as snn

    x, w, b = _rand((2, 9, 11, ic), 1), _rand((oc, ic, k, k), 2, 0.3), _rand((oc,), 3, 0.1)
    bnp = _bn(oc, 4) if bn else None
    y, desc = _run(ctx, snn.deconv2d_plan(ctx, 2, 9, 11, w, b, stride=s, same=same, act=act, leaky=0.2, bn=bnp), x)
    np.testing.assert_allclose(y, O.deconv2d(x, w, b, s, same, act, 0.2, bnp), err_msg=desc, **TOL)
    if k == 4 and s == 2 and same and not act and not bn:
        np.testing.assert_allclose(y[0], O.deconv4x4s2_shader(x[0], w, b), **TOL)


@pytest.mark.gpu
def test_deconv2d_k4s2_matches_reference_shader_restated(ctx):
    import shadernn_amd as snn

    x, w, b = _rand((1, 12, 10, 8), 5), _rand((8, 8, 4, 4), 6, 0.3), _rand((8,), 7, 0.1)
    y, desc = _run(ctx, snn.deconv2d_plan(ctx, 1, 12, 10, w, b, stride=2, same=True), x)
    np.testing.assert_allclose(y[0], O.deconv4x4s2_shader(x[0], w, b), err_msg=desc, **TOL)
