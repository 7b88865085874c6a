"""Randomised parity sweep of the convolution entry point (fixed seed): shapes, strides, kernel sizes, padding modes, activations, BN, dtype,
batch -- every routing decision of snnhip_conv2d_plan_create (generic / thin / MFMA with its tap-pair, static-tap, narrow-chunk, split-K and
LDS-epilogue variants) against the CPU oracle."""
import os

import numpy as np
import pytest

import oracle_lib as O
from test_ops_gpu import TOL, _bn, _rand

pytestmark = pytest.mark.gpu

ACTS = ["", "relu", "relu6", "tanh", "sigmoid", "leakyRelu", "SiLU"]
PADS = ["constant", "replicate", "reflect"]


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        k = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 7, 9]))
        s = int(rng.choice([1, 1, 1, 2]))
        ic = int(rng.choice([1, 3, 4, 8, 12, 16, 24, 32, 48, 64, 96, 130]))
        oc = int(rng.choice([1, 3, 4, 8, 16, 24, 32, 40, 64, 96, 128, 160]))
        h, w = int(rng.integers(max(k, 5), 40)), int(rng.integers(max(k, 5), 44))
        b = int(rng.choice([1, 1, 2, 3]))
        if b * h * w * ic * k * k * oc > 3e8:  # keep the oracle fast
            continue
        padding = str(rng.choice(["same", "valid"]))
        if k % 2 == 0 and padding == "valid":  # the reference's size rule underflows there (see test_even_kernel_valid_padding_is_refused)
            padding = "same"
        out.append((b, h, w, ic, oc, k, s, str(rng.choice(ACTS)), str(rng.choice(PADS)), bool(rng.integers(0, 2)), padding,
                    int(rng.integers(0, 1 << 30))))
    return out


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("case", _cases(int(os.environ.get("SNN_FUZZ_CASES", "60")), int(os.environ.get("SNN_FUZZ_SEED", "20260927"))), ids=lambda c: "n%d_%dx%d_%d-%d_k%ds%d_%s_%s_bn%d_%s" % c[:11])
def test_conv_random_shapes_match_oracle(ctx, case, dtype):
    import shadernn_amd as snn

    n, h, w, ic, oc, k, s, act, pad_mode, use_bn, padding, seed = case
    x = _rand((n, h, w, ic), seed)
    wt = _rand((oc, ic, k, k), seed + 1, 1.0 / np.sqrt(ic * k * k))
    b = _rand((oc,), seed + 2, 0.2)
    bn = _bn(oc, seed + 3) if use_bn else None
    pads = O.padding_offsets(padding, k)
    dt = snn.F16 if dtype == "f16" else snn.F32
    plan = snn.conv2d_plan(ctx, n, h, w, wt, b, stride=s, pads=pads, pad_mode=pad_mode, act=act, leaky=0.15, bn=bn, dtype=dt)
    xt = snn.Tensor.from_numpy(ctx, x, dtype=dt)
    yt = plan(xt)
    y, desc = yt.numpy(), plan.describe()
    xt.free(), yt.free(), plan.destroy()
    if dtype == "f16":
        want = O._h(O.conv2d(O._h(x), O._h(wt), b, s, pads, pad_mode, act, 0.15, bn))
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(y / scale, want / scale, err_msg=desc, rtol=3e-3, atol=3e-3)
    else:
        want = O.conv2d(x, wt, b, s, pads, pad_mode, act, 0.15, bn)
        assert y.shape == want.shape, desc
        np.testing.assert_allclose(y, want, err_msg=desc, **TOL)


def test_even_kernel_valid_padding_is_refused(ctx):
    """Conv2DLayer::getOutputScaleDimAdjustment computes `offset[0] + offset[1] - 1` in uint32 for even kernels (conv2d.cpp:110): with zero
    padding that wraps to 4294967295 and the reference's own output extent is garbage.  The restated rule reproduces the wrap; the plan is
    refused instead of allocating it."""
    import shadernn_amd as snn

    with pytest.raises(snn.SnnHipError):
        snn.conv2d_plan(ctx, 1, 12, 12, _rand((8, 8, 4, 4), 1), None, stride=1, pads=(0, 0, 0, 0))
