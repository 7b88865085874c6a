"""The launch trace behind bench.py's per-kernel roofline (snnhip_trace_begin / _end / _report, include/snnhip.h): every kernel a plan launches is
stamped with its own dispatch start / end and booked on the plan that launched it, grouped by kernel function."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_trace_books_each_launch_on_its_kernel_function(ctx):
    import shadernn_amd as snn
    from shadernn_amd import capi, models

    net = models.espcn_weights(seed=1)
    H, W = 72, 96
    x = np.random.default_rng(3).random((1, H, W, 1), dtype=np.float32)
    r = snn.EspcnRunner(ctx, net, 1, H, W, fused=True)
    y0 = r(x)
    capi.trace_begin()
    for _ in range(5):
        r.run_device()
    ctx.sync()
    t = capi.trace_end()
    assert t["launches"] == 10                       # two fused kernels per inference
    ks = {k["function"]: k for k in t["kernels"]}
    assert set(ks) == {"conv_kxk_c1o16_wino3x3_c16o16_kernel", "conv3x3_c16o4_d2s_tanh_kernel"}
    a, b = ks["conv_kxk_c1o16_wino3x3_c16o16_kernel"], ks["conv3x3_c16o4_d2s_tanh_kernel"]
    px = H * W
    assert a["launches"] == 5 and a["main_launches"] == 5 and b["launches"] == 5
    assert a["flops"] == pytest.approx(5 * (800 + 4608) * px) and b["flops"] == pytest.approx(5 * 1152 * px)
    assert a["bytes"] == pytest.approx(5 * 4.0 * (px * 17 + 16 * 25 + 16 * 16 * 9))   # the fused launch's own traffic: x once, the 16-channel tensor once, weights
    assert 0 < a["total_ms"] < 50 and 0 < b["total_ms"] < 50
    assert "conv_kxk_c1o16_wino3x3_c16o16_kernel<5" in a["instances"][0]["name"].replace(" ", "")
    assert any("winograd" in p for p in a["instances"][0]["plans"])
    # tracing changes nothing about the result, and it is off again afterwards
    np.testing.assert_array_equal(r(x), y0)
    capi.trace_begin()
    t2 = capi.trace_end()
    assert t2["launches"] == 0 and t2["kernels"] == []


def test_trace_splits_a_multi_launch_plan_by_kernel(ctx):
    """An InstanceNorm is three launches (statistics sweep, fold, normalise sweep): three records with their own bytes, never one 'step'."""
    import shadernn_amd as snn
    from shadernn_amd import capi

    N, H, W, C = 2, 40, 56, 32
    x = np.random.default_rng(4).standard_normal((N, H, W, C)).astype(np.float32)
    tx, ty = snn.Tensor(ctx, N, H, W, C), snn.Tensor(ctx, N, H, W, C)
    tx.upload(x)
    p = snn.instancenorm_plan(ctx, N, H, W, C, np.zeros(C, np.float32), np.ones(C, np.float32), act="")
    p.run(tx, ty)
    ctx.sync()
    capi.trace_begin()
    p.run(tx, ty)
    ctx.sync()
    t = capi.trace_end()
    assert t["launches"] == 3
    byf = {k["function"]: k for k in t["kernels"]}
    assert set(byf) == {"instancenorm_kernel", "instancenorm_fold_kernel"}
    sweep = byf["instancenorm_kernel"]
    assert sweep["launches"] == 2 and len(sweep["instances"]) == 2          # statistics and normalise are two instantiations of one function
    tensor = 4.0 * N * H * W * C
    assert sweep["bytes"] == pytest.approx(3 * tensor)                       # read once for the statistics, read + written by the normalise sweep
    assert byf["instancenorm_fold_kernel"]["bytes"] == 0
