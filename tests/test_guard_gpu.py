"""Device-side guard mode (SNNHIP_GUARD=1, include/snnhip.h): red zones of 0xFF bytes around every device allocation of the library, verified by
snnhip_sync / snnhip_tensor_download.  The mode is fixed when the library makes its first allocation, so these tests run the guarded library in a
child process.  SURVEY section 5 ("compute-sanitizer equivalent"); the reference has no device-side checking (core/CMakeLists.txt:235)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _guarded(code, timeout=600):
    env = dict(os.environ, SNNHIP_GUARD="1", PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout, env=env, cwd=ROOT)
    return r.returncode, r.stdout


def test_guard_catches_a_kernel_that_writes_outside_its_tensor(ctx):
    rc, out = _guarded("""
        import numpy as np
        import shadernn_amd as snn
        from shadernn_amd import capi
        snn.load_library()
        assert capi.lib().snnhip_guard_active() == 1
        ctx = snn.Context(0)
        t = snn.Tensor.from_numpy(ctx, np.ones((1, 5, 7, 3), np.float32))      # 420 bytes: the region ends 12 bytes short of a 16-byte boundary
        ctx.sync()                                                                 # clean
        for off, where in ((0, "past the END"), (11, "past the END"), (4000, "past the END"), (-1, "in FRONT"), (-65536, "in FRONT")):
            capi.check(capi.lib().snnhip_guard_selftest(ctx.h, t.h, off))         # one byte from a kernel, outside the tensor
            try:
                ctx.sync()
            except snn.SnnHipError as e:
                assert e.code == capi.E_GUARD and where in str(e) and "tensor 1x5x7x3" in str(e), str(e)
            else:
                raise AssertionError("offset %d not caught" % off)
            ctx.sync()                                                             # the damaged zone was re-poisoned: the report is not sticky
        np.testing.assert_array_equal(t.numpy(), 1.0)                              # download checks too, and is clean
        print("GUARD-OK")
    """)
    assert rc == 0 and "GUARD-OK" in out, out[-3000:]


def test_fresh_tensors_and_scratch_read_as_nan_under_the_guard(ctx):
    rc, out = _guarded("""
        import numpy as np
        import shadernn_amd as snn
        snn.load_library()
        ctx = snn.Context(0)
        t = snn.Tensor(ctx, 1, 4, 4, 8)
        assert np.isnan(t.numpy()).all()       # 0xFFFFFFFF: nothing reads as a plausible number before it is written
        h = snn.Tensor(ctx, 1, 4, 4, 8, dtype=snn.F16)
        assert np.isnan(h.numpy()).all()       # 0xFFFF
        print("GUARD-OK")
    """)
    assert rc == 0 and "GUARD-OK" in out, out[-3000:]


def test_the_hot_path_is_clean_under_the_guard(ctx):
    """ESPCN fused and layer by layer against the oracle with every allocation between red zones (tools/sanitize.sh guard runs the fuzz, inverted-residual
    and bench-size tests the same way)."""
    rc, out = _guarded("""
        import numpy as np
        import shadernn_amd as snn
        from shadernn_amd import models
        import oracle_lib as O
        snn.load_library()
        ctx = snn.Context(0)
        net = models.espcn_weights(seed=1)
        x = np.random.default_rng(3).random((1, 90, 130, 1), dtype=np.float32)
        want = O.espcn_forward(net, x)
        for fused in (False, True):
            y = snn.EspcnRunner(ctx, net, 1, 90, 130, fused=fused)(x)
            np.testing.assert_allclose(y, want, rtol=1e-4, atol=1e-4)
        ctx.sync()
        print("GUARD-OK")
    """)
    assert rc == 0 and "GUARD-OK" in out, out[-3000:]


def test_guard_catches_a_deliberately_broken_convolution_build(ctx):
    """build/abl/libsnnhip_guardbreak.so (tools/sanitize.sh guard builds it: conv2d_generic.hip with -DSNNHIP_GUARD_BREAK, an off-by-one output row bound):
    on a 20-row output whose last 8-row tile is half empty the kernel stores a 21st row behind the tensor -- SNNHIP_E_GUARD names the output tensor."""
    lib = os.path.join(ROOT, "build", "abl", "libsnnhip_guardbreak.so")
    if not os.path.exists(lib):
        pytest.skip("tools/sanitize.sh guard builds the broken library")
    code = """
        import numpy as np
        import shadernn_amd as snn
        from shadernn_amd import capi
        snn.load_library()
        ctx = snn.Context(0)
        rng = np.random.default_rng(1)
        w = rng.standard_normal((8, 4, 3, 3)).astype(np.float32)
        x = snn.Tensor.from_numpy(ctx, rng.random((1, 20, 33, 4), dtype=np.float32))
        plan = snn.conv2d_plan(ctx, 1, 20, 33, w, np.zeros(8, np.float32), stride=1, pads=snn.same_padding(3), act="relu")
        assert "generic" in plan.describe(), plan.describe()
        y = plan(x)
        try:
            ctx.sync()
        except snn.SnnHipError as e:
            assert e.code == capi.E_GUARD and "past the END" in str(e) and "tensor 1x20x33x8" in str(e), str(e)
            print("GUARD-CAUGHT")
        else:
            print("GUARD-MISSED")
    """
    env_lib = dict(SNNHIP_LIB_PATH=lib, SNNHIP_CONV="generic")
    os.environ.update(env_lib)
    try:
        rc, out = _guarded(code)
    finally:
        for k in env_lib:
            os.environ.pop(k, None)
    assert rc == 0 and "GUARD-CAUGHT" in out, out[-3000:]
    rc, out = _guarded("import os; os.environ['SNNHIP_CONV'] = 'generic'\n" + textwrap.dedent(code))  # the product library, same layer: clean
    assert rc == 0 and "GUARD-MISSED" in out, out[-3000:]
