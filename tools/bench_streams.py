#!/usr/bin/env python
"""bench_streams.py -- the same batch of a graph-shaped model as ONE runner on one stream vs S runners of batch/S on S HIP streams
(independent images, no cross-stream dependency): kernels of different streams fill each other's launch tails.  Wall-clock timing
(host clock around `reps` rounds, device synchronised on both sides).

    python tools/bench_streams.py --model mobilenetv2 --batch 32 --streams 2 [--fp16]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mobilenetv2")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--fp16", action="store_true")
    ap.add_argument("--size", type=int, default=224)
    args = ap.parse_args()
    import torch

    import shadernn_amd as snn
    from shadernn_amd import models

    snn.load_library()
    make = {"mobilenetv2": lambda: models.mobilenetv2(seed=1), "resnet18": lambda: models.resnet18(seed=1)}[args.model]
    dt = snn.F16 if args.fp16 else snn.F32
    for S in sorted({1, args.streams}):
        streams = [torch.cuda.Stream() for _ in range(S)]
        ctxs = [snn.Context(0, stream=s.cuda_stream) for s in streams]
        runners = [snn.GraphRunner(c, make(), args.batch // S, args.size, args.size, dtype=dt) for c in ctxs]
        graphs = []
        for c, r in zip(ctxs, runners):
            r.x.upload(np.random.default_rng(1).random(r.in_shape, dtype=np.float32))
            for _ in range(2):
                r.run_device()
            c.sync()
            with snn.Graph.capture(c) as g:  # one host call per stream and round
                r.run_device()
            graphs.append(g)
        for g in graphs:
            g.launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            for g in graphs:
                g.launch()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.reps
        print("%s %s batch %d on %d stream(s): %.3f ms per batch, %.0f images/s" % (args.model, "fp16" if args.fp16 else "fp32", args.batch, S, ms, args.batch / ms * 1e3), flush=True)


if __name__ == "__main__":
    main()
