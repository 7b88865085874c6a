#!/usr/bin/env python
"""Times the UpSampling x2 -> reflect Pad(1) -> Conv2D 3x3 chain (Candy's two up-convolutions at micro-batch 16) stand-alone: one chain plan, events
around `reps` launches.   python tools/bench_upconv.py [--reps 20] [--stats] [--n 16] [--segs 0,2,3,4,6,8 --rounds 2]
(--segs: sweep SNNHIP_UPCONV_SEGS, 0 = the planner's own choice; the candidates alternate over `rounds` passes so that box drift hits all alike)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shadernn_amd as snn

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--n", type=int, default=16)
ap.add_argument("--segs", default="0")
ap.add_argument("--rounds", type=int, default=1)
ap.add_argument("--stats", action="store_true", help="with the InstanceNorm behind the convolution (rule F: block statistics + fold)")
a = ap.parse_args()
ctx = snn.Context(0)
rng = np.random.default_rng(1)
segs = [int(v) for v in a.segs.split(",")]
for (n, h, w, ic, oc) in [(a.n, 408, 688, 64, 32), (a.n, 203, 343, 128, 64)]:  # Candy 720p: the maps grow by 2 per reflect Pad -> Conv2D (size rule Q20)
    x = rng.random((n, h, w, ic), dtype=np.float32)
    wt = (rng.standard_normal((oc, ic, 3, 3)) / np.sqrt(9 * ic)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, oc).astype(np.float32)
    plans = [snn.upsample_plan(ctx, n, h, w, ic, 2.0, "nearest"), snn.pad_plan(ctx, n, 2 * h, 2 * w, ic, (1, 1, 1, 1), "reflect"),
             snn.conv2d_plan(ctx, n, 2 * h + 2, 2 * w + 2, wt, b, stride=1, pads=(0, 0, 0, 0), act="", dtype=snn.F16)]
    if a.stats:
        plans.append(snn.instancenorm_plan(ctx, n, 2 * h + 2, 2 * w + 2, oc, np.zeros(oc, np.float32), np.ones(oc, np.float32), act="relu"))
    xt = snn.Tensor.from_numpy(ctx, x, dtype=snn.F16)
    chains = {}
    for sg in segs:
        snn.set_option("SNNHIP_UPCONV_SEGS", sg if sg else None)
        chains[sg] = snn.chain_plan(ctx, plans)
    snn.set_option("SNNHIP_UPCONV_SEGS", None)
    yt = snn.Tensor(ctx, *chains[segs[0]].out_shape(), dtype=snn.F16)
    best = {sg: 1e30 for sg in segs}
    for _ in range(a.rounds):
        for sg in segs:
            chain = chains[sg]
            for _ in range(3):
                chain.run(xt, yt)
            ctx.sync()
            t = snn.Timer(ctx)
            t.start()
            for _ in range(a.reps):
                chain.run(xt, yt)
            t.stop()
            ctx.sync()
            best[sg] = min(best[sg], 1e3 * t.elapsed_ms() / a.reps)
    for sg in segs:
        d = chains[sg].describe()
        print("%d x %dx%d %d->%d segs %s: %.1f us | %s" % (n, h, w, ic, oc, sg or "auto", best[sg], d[d.find("segments="):][:28] if len(segs) > 1 else d[:260]), flush=True)
