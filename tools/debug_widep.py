"""debug: the persistent wide kernel against conv2d_wide_kernel on one small layer, mismatches located by tile / row / column / channel"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import shadernn_amd as snn
import oracle_lib as O
from test_ops_gpu import _bn, _rand
snn.load_library()
ctx = snn.Context(0)
os.environ["SNNHIP_CONV"] = "wide"
N, H, W, C = int(os.environ.get("DBG_N", 1)), int(os.environ.get("DBG_H", 37)), int(os.environ.get("DBG_W", 70)), 128
x = _rand((N, H, W, C), 81); w = _rand((C, C, 3, 3), 82, 1.0 / np.sqrt(C * 9)); b = _rand((C,), 83, 0.1); bn = _bn(C, 84)
pads = O.padding_offsets("same", 3)
def run(pad_mode, act, bnp, persist):
    os.environ["SNNHIP_WIDE_PERSIST"] = "1" if persist else "0"
    plan = snn.conv2d_plan(ctx, N, H, W, w, b, stride=1, pads=pads, pad_mode=pad_mode, act=act, bn=bnp, dtype=snn.F16)
    y, d = plan(snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)).numpy(), plan.describe()
    plan.destroy()
    return y, d
for pad_mode, act, use_bn in (("constant", "relu", False), ("constant", "relu", True), ("replicate", "", True), ("reflect", "", False)):
    for rep in range(2):
        y, d = run(pad_mode, act, bn if use_bn else None, True)
        y0, d0 = run(pad_mode, act, bn if use_bn else None, False)
        bad = ~np.isclose(y, y0, rtol=1e-3, atol=1e-3) | np.isnan(y)
        print("== %s act=%r bn=%s rep %d: nan %d mismatches %d of %d   [%s]" % (pad_mode, act, use_bn, rep, int(np.isnan(y).sum()), int(bad.sum()), y.size, "persistent" in d))
        if bad.any():
            idx = np.argwhere(bad)
            print("   images", sorted(set(idx[:, 0].tolist())), "tile rows", sorted(set((idx[:, 1] // 8).tolist())), "tile cols", sorted(set((idx[:, 2] // 32).tolist())))
            print("   rows%8", np.bincount(idx[:, 1] % 8, minlength=8).tolist())
            print("   cols%32", np.bincount(idx[:, 2] % 32, minlength=32).tolist())
            print("   ch//8", np.bincount(idx[:, 3] // 8, minlength=16).tolist())
            print("   ch%8", np.bincount(idx[:, 3] % 8, minlength=8).tolist())
            i = tuple(idx[0]); print("   first", i, float(y[i]), float(y0[i]))
