#!/usr/bin/env python
"""Times a BASELINE config through snn_pool (include/snn_c.h) with R replicas on ONE GPU: tools/bench_pool.py c5 1 2 [--steps 3]
(R > 1 = R host threads / contexts / streams on device 0, the batch split between them: do two streams fill each other's launch gaps and tails?)"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("replicas", type=int, nargs="+")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="global batch (default: the config's); e.g. c2 with --batch 2: two frames per step, one per replica")
    a = ap.parse_args()
    import bench
    from shadernn_amd import host, models

    cfg = bench.CONFIGS[a.config]
    net = bench.make_net(a.config)
    H, W = cfg["hw"]
    tmp = tempfile.mkdtemp(prefix="snn_pool_")
    path = models.write_json(net, W, H, os.path.join(tmp, "m.json"), bin_weights=True)
    B = a.batch or cfg.get("global", cfg.get("per_rank"))
    x = np.random.default_rng(7767517).random((B, H, W, cfg["cin"]), dtype=np.float32)
    ref = None
    for r in a.replicas:
        pool = host.Pool(path, W, H, cfg["cin"], devices=[0] * r, global_batch=B, micro_batch=min(cfg["micro"], B // r), prefer_half=cfg["dtype"] == "f16")
        pool.upload(x)
        pool.run(2)
        ts = [pool.run(a.steps) / a.steps for _ in range(3)]
        y = pool.output()
        if ref is None:
            ref = y
        print("%s replicas=%d: %.4f ms/step (median of 3 x %d steps; min %.4f) %.1f images/s | max |y - y_1replica| %.3g" % (
            a.config, r, 1e3 * float(np.median(ts)), a.steps, 1e3 * min(ts), B / float(np.median(ts)), float(np.abs(y - ref).max())))
        pool.close()


if __name__ == "__main__":
    main()
