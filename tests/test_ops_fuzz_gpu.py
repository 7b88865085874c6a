"""Randomised parity sweep (fixed seed) of the non-conv operators through the C-ABI vs the CPU oracle: depthwise, pooling, pad, upsampling,
instance norm, add with inputs of different extent, transposed convolution -- fp32 and fp16 tensors."""
import os

import numpy as np
import pytest

import oracle_lib as O
from test_ops_gpu import TOL, _bn, _rand

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("SNN_FUZZ_CASES", "25"))
SEED = int(os.environ.get("SNN_FUZZ_SEED", "20260927"))
H16 = dict(rtol=3e-3, atol=3e-3)


def _run(ctx, plan, xs, dt):
    import shadernn_amd as snn

    ts = [snn.Tensor.from_numpy(ctx, x, dtype=dt) for x in xs]
    yt = snn.Tensor(ctx, *plan.out_shape(), dtype=dt)
    plan.run(ts if len(ts) > 1 else ts[0], yt)
    y, desc = yt.numpy(), plan.describe()
    for t in ts + [yt]:
        t.free()
    plan.destroy()
    return y, desc


def _q(a, f16):
    return O._h(a) if f16 else a


@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16"])
@pytest.mark.parametrize("i", range(N_CASES))
def test_depthwise_random(ctx, i, f16):
    import shadernn_amd as snn

    rng = np.random.default_rng(SEED + i)
    c, k, s = int(rng.choice([3, 4, 8, 12, 32, 96, 144])), int(rng.choice([1, 3, 3, 5])), int(rng.choice([1, 2]))
    n, h, w = int(rng.choice([1, 2])), int(rng.integers(k + 1, 30)), int(rng.integers(k + 1, 33))
    act = str(rng.choice(["", "relu", "relu6", "leakyRelu", "tanh"]))
    x, wt, b = _rand((n, h, w, c), i), _rand((c, k, k), i + 1, 0.4), _rand((c,), i + 2, 0.1)
    bn = _bn(c, i + 3) if rng.integers(0, 2) else None
    pads = O.padding_offsets("same", k)
    dt = snn.F16 if f16 else snn.F32
    y, desc = _run(ctx, snn.conv2d_plan(ctx, n, h, w, wt, b, stride=s, pads=pads, act=act, leaky=0.1, bn=bn, depthwise=True, dtype=dt), [x], dt)
    want = _q(O.depthwise(_q(x, f16), wt, b, s, pads, act, 0.1, bn), f16)
    np.testing.assert_allclose(y, want, err_msg=desc, **(H16 if f16 else TOL))


@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16"])
@pytest.mark.parametrize("i", range(N_CASES))
def test_pool_pad_upsample_random(ctx, i, f16):
    import shadernn_amd as snn

    rng = np.random.default_rng(SEED + 1000 + i)
    n, h, w, c = int(rng.choice([1, 2])), int(rng.integers(5, 28)), int(rng.integers(5, 31)), int(rng.choice([3, 4, 7, 16, 40]))
    x = _rand((n, h, w, c), i, 2.0)
    dt = snn.F16 if f16 else snn.F32
    tol = H16 if f16 else TOL
    k, s = int(rng.choice([2, 3])), int(rng.choice([1, 2]))
    kind, same = str(rng.choice(["max", "avg"])), bool(rng.integers(0, 2))
    y, desc = _run(ctx, snn.pool2d_plan(ctx, n, h, w, c, k, s, kind=kind, same=same), [x], dt)
    np.testing.assert_allclose(y, _q(O.pool2d(_q(x, f16), k, s, kind, same), f16), err_msg=desc, **tol)
    pads = tuple(int(v) for v in rng.integers(0, 4, 4))
    mode = str(rng.choice(["constant", "replicate", "reflect"]))
    y, desc = _run(ctx, snn.pad_plan(ctx, n, h, w, c, pads, mode), [x], dt)
    np.testing.assert_array_equal(y, O.pad(_q(x, f16), pads, mode), err_msg=desc)
    scale, interp = float(rng.choice([2.0, 3.0, 1.5])), str(rng.choice(["nearest", "bilinear"]))
    y, desc = _run(ctx, snn.upsample_plan(ctx, n, h, w, c, scale, interp), [x], dt)
    np.testing.assert_allclose(y, _q(O.upsample(_q(x, f16), scale, interp), f16), err_msg=desc, **tol)


@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16"])
@pytest.mark.parametrize("i", range(N_CASES))
def test_instancenorm_add_deconv_random(ctx, i, f16):
    import shadernn_amd as snn

    rng = np.random.default_rng(SEED + 2000 + i)
    n, h, w, c = int(rng.choice([1, 2])), int(rng.integers(4, 40)), int(rng.integers(4, 45)), int(rng.choice([3, 4, 8, 32, 128]))
    x = _rand((n, h, w, c), i, 1.5) + 0.3
    dt = snn.F16 if f16 else snn.F32
    tol = H16 if f16 else TOL
    beta, gamma = _rand((c,), i + 1, 0.2), _rand((c,), i + 2, 0.5) + 1.0
    act = str(rng.choice(["", "relu"]))
    y, desc = _run(ctx, snn.instancenorm_plan(ctx, n, h, w, c, beta, gamma, act=act), [x], dt)
    want = _q(O.instancenorm(_q(x, f16), beta, gamma, act), f16)
    np.testing.assert_allclose(y, want, err_msg=desc, **(dict(rtol=1e-2, atol=1e-2) if f16 else dict(rtol=1e-4, atol=2e-4)))
    # Add with a second input that is smaller by up to 4 pixels (the reference's max-extent rule)
    dh, dw = int(rng.integers(0, min(5, h))), int(rng.integers(0, min(5, w)))
    b2 = _rand((n, h - dh, w - dw, c), i + 3)
    y, desc = _run(ctx, snn.add_plan(ctx, n, h, w, c, act="relu"), [x, b2], dt)
    np.testing.assert_allclose(y, _q(O.add_act(_q(x, f16), _q(b2, f16), "relu", 0.0), f16), err_msg=desc, **tol)
    if c <= 32 and h * w <= 400:
        k, s = (4, 2) if rng.integers(0, 2) else (3, 1)
        oc = int(rng.choice([2, 4, 9]))
        wt, bb = _rand((oc, c, k, k), i + 4, 0.3), _rand((oc,), i + 5, 0.1)
        y, desc = _run(ctx, snn.deconv2d_plan(ctx, n, h, w, wt, bb, stride=s, same=True, act="relu"), [x], dt)
        np.testing.assert_allclose(y, _q(O.deconv2d(_q(x, f16), wt, bb, s, True, "relu"), f16), err_msg=desc, **tol)
