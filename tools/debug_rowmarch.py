"""Probe inputs for the row-marching kernel (run on the GPU box): delta weights + ramp inputs show which pixel / tap lands where."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shadernn_amd as snn

np.set_printoptions(linewidth=250, precision=2, suppress=True)
ctx = snn.Context(0)
K, IC, OC, H, W = 9, 16, 1, 12, 20


def run(x, w, tile=False):
    if tile:
        os.environ["SNNHIP_ROWFOLD"] = "tile"
    else:
        os.environ.pop("SNNHIP_ROWFOLD", None)
    plan = snn.conv2d_plan(ctx, 1, H, W, w, np.zeros(OC, np.float32), stride=1, pads=(4, 4, 4, 4), pad_mode="constant", act="", dtype=snn.F16)
    y = plan(snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)).numpy()
    return y[0, :, :, 0], plan.describe()


def delta(fy, fx, ic=0):
    w = np.zeros((OC, IC, K, K), np.float32)
    w[0, ic, fy, fx] = 1.0
    return w


ones = np.ones((1, H, W, IC), np.float32)
colramp = np.broadcast_to(np.arange(W, dtype=np.float32)[None, None, :, None], (1, H, W, IC)).copy()
rowramp = np.broadcast_to(np.arange(H, dtype=np.float32)[None, :, None, None], (1, H, W, IC)).copy()
for name, x, w in [("ones, centre tap", ones, delta(4, 4)), ("col ramp, centre tap", colramp, delta(4, 4)), ("row ramp, centre tap", rowramp, delta(4, 4)),
                   ("col ramp, tap (4,0)", colramp, delta(4, 0)), ("col ramp, tap (4,8)", colramp, delta(4, 8)), ("row ramp, tap (0,4)", rowramp, delta(0, 4)),
                   ("col ramp, centre tap, ic=9", colramp, delta(4, 4, 9))]:
    y, d = run(x, w)
    yt, _ = run(x, w, tile=True)
    print("====", name, "|", d[:120])
    print("march:\n", y[:4])
    print("tile:\n", yt[:4])
    print("max diff", np.abs(y - yt).max())
