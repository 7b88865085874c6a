"""debug_split.py -- developer script: where the split-precision irb kernels differ from the oracle (per channel / row / column / image)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import oracle_lib as O
from test_ops_gpu import _rand
from test_irb_gpu import _layers
import shadernn_amd as snn

os.environ["SNNHIP_IRB_BAND"] = "1"
case = tuple(int(v) for v in sys.argv[1].split(",")) + (False, ("relu6", "relu6", ""))
if len(sys.argv) > 2 and sys.argv[2] != "-":
    os.environ["SNNHIP_IRB_BAND_GEOM"] = sys.argv[2]
N, H, W, C, Ch, Co, s, res, acts = case
ctx = snn.Context(0)
x = _rand((N, H, W, C), 271)
(we, be, bne), (wd, bd, bnd), (wp, bp, bnp) = _layers(case, 280)
h = O.conv2d(x, we, be, 1, (0, 0, 0, 0), "constant", acts[0], 0.1, bne, threads=8)
dd = O.depthwise(h, wd, bd, s, O.padding_offsets("same", 3), acts[1], 0.1, bnd)
want = O.conv2d(dd, wp, bp, 1, (0, 0, 0, 0), "constant", acts[2], 0.1, bnp, threads=8)
for split in ("0", "1"):
    os.environ["SNNHIP_IRB_SPLIT"] = split
    pe = snn.conv2d_plan(ctx, N, H, W, we, be, act=acts[0], leaky=0.1, bn=bne)
    pd = snn.conv2d_plan(ctx, N, H, W, wd, bd, stride=s, pads=O.padding_offsets("same", 3), act=acts[1], leaky=0.1, bn=bnd, depthwise=True)
    _, OH, OW, _ = pd.out_shape()
    pp = snn.conv2d_plan(ctx, N, OH, OW, wp, bp, act=acts[2], leaky=0.1, bn=bnp)
    plan = snn.chain_plan(ctx, [pe, pd, pp])
    got = plan(snn.Tensor.from_numpy(ctx, x)).numpy()
    e = np.abs(got - want)
    print("split", split, plan.describe()[:200])
    print("  max err %.3g, mean %.3g, nan %d" % (np.nanmax(e), np.nanmean(e), np.isnan(got).sum()))
    print("  per image", e.max(axis=(1, 2, 3)))
    print("  per row  ", np.round(e.max(axis=(0, 2, 3)), 4)[:64])
    print("  per col  ", np.round(e.max(axis=(0, 1, 3)), 4)[:64])
    print("  per chan ", np.round(e.max(axis=(0, 1, 2)), 4))
