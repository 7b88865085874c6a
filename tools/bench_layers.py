#!/usr/bin/env python
"""bench_layers.py -- single-layer throughput of the HIP conv kernels at the BASELINE shapes (C1, ResNet-18 b32 layers,
MobileNetV2 pointwise/depthwise b32), each with its algorithmic FLOPs / bytes and the fraction of the bounding roofline
(fp32 MFMA 157.3 TFLOP/s, HBM 8 TB/s).  Timing: `reps` back-to-back launches between two HIP events on the context stream.

    python tools/bench_layers.py [--reps 50] [--force mfma|generic] [--json out.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK_TF, PEAK_GBS = 157.3, 8000.0

LAYERS = [
    # name, N, H, W, IC, OC, k, stride, depthwise
    ("C1 conv3x3 3->64 @224 b1", 1, 224, 224, 3, 64, 3, 1, False),
    ("resnet stem 7x7s2 3->64 @224 b32", 32, 224, 224, 3, 64, 7, 2, False),
    ("resnet l1 3x3 64->64 @56 b32", 32, 56, 56, 64, 64, 3, 1, False),
    ("resnet l2a 3x3s2 64->128 @56 b32", 32, 56, 56, 64, 128, 3, 2, False),
    ("resnet l2 3x3 128->128 @28 b32", 32, 28, 28, 128, 128, 3, 1, False),
    ("resnet l2 ds 1x1s2 64->128 @56 b32", 32, 56, 56, 64, 128, 1, 2, False),
    ("resnet l3 3x3 256->256 @14 b32", 32, 14, 14, 256, 256, 3, 1, False),
    ("resnet l4 3x3 512->512 @7 b32", 32, 7, 7, 512, 512, 3, 1, False),
    ("mbv2 pw 32->16 @112 b32", 32, 112, 112, 32, 16, 1, 1, False),
    ("mbv2 pw 16->96 @112 b32", 32, 112, 112, 16, 96, 1, 1, False),
    ("mbv2 pw 144->24 @56 b32", 32, 56, 56, 144, 24, 1, 1, False),
    ("mbv2 pw 192->32 @28 b32", 32, 28, 28, 192, 32, 1, 1, False),
    ("mbv2 pw 384->64 @14 b32", 32, 14, 14, 384, 64, 1, 1, False),
    ("mbv2 pw 960->320 @7 b32", 32, 7, 7, 960, 320, 1, 1, False),
    ("mbv2 pw 320->1280 @7 b32", 32, 7, 7, 320, 1280, 1, 1, False),
    ("mbv2 dw 3x3s2 96 @112 b32", 32, 112, 112, 96, 96, 3, 2, True),
    ("mbv2 dw 3x3 144 @56 b32", 32, 56, 56, 144, 144, 3, 1, True),
    ("mbv2 dw 3x3 384 @14 b32", 32, 14, 14, 384, 384, 3, 1, True),
    ("unet 3x3 128->128 @128 b8", 8, 128, 128, 128, 128, 3, 1, False),
    ("unet 3x3 64->64 @256 b8", 8, 256, 256, 64, 64, 3, 1, False),
    ("espcn conv2 3x3 16->16 @1080p b1", 1, 1080, 1920, 16, 16, 3, 1, False),
    ("espcn conv3 3x3 16->4 @1080p b1", 1, 1080, 1920, 16, 4, 3, 1, False),
    ("candy out 9x9 32->3 @720p b1", 1, 728, 1288, 32, 3, 9, 1, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--force", default=None)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None)
    ap.add_argument("--fp16", action="store_true", help="half tensors (fp16 MFMA path; roofline against 2516 TFLOP/s)")
    ap.add_argument("--shape", action="append", default=[], help="extra ad-hoc layer N,H,W,IC,OC,k,stride (repeatable; with --only=adhoc runs just these)")
    args = ap.parse_args()
    global PEAK_TF
    if args.fp16:
        PEAK_TF = 2516.6
    for sh in args.shape:
        v = [int(t) for t in sh.split(",")]
        LAYERS.append(("adhoc %dx%d s%d %d->%d @%dx%d b%d" % (v[5], v[5], v[6], v[3], v[4], v[1], v[2], v[0]), v[0], v[1], v[2], v[3], v[4], v[5], v[6], False))
    if args.force:
        os.environ["SNNHIP_CONV"] = args.force
    import shadernn_amd as snn

    snn.load_library()
    ctx = snn.Context(0)
    rng = np.random.default_rng(1)
    rows = []
    for (name, N, H, W, IC, OC, k, s, dw) in LAYERS:
        if args.only and args.only not in name:
            continue
        x = rng.random((N, H, W, IC), dtype=np.float32)
        w = (rng.standard_normal((OC, k, k) if dw else (OC, IC, k, k)) / np.sqrt(k * k * (1 if dw else IC))).astype(np.float32)
        b = rng.uniform(-0.1, 0.1, OC).astype(np.float32)
        dt = snn.F16 if args.fp16 else snn.F32
        plan = snn.conv2d_plan(ctx, N, H, W, w, b, stride=s, pads=snn.same_padding(k), pad_mode="constant", act="relu", depthwise=dw, dtype=dt)
        xt = snn.Tensor.from_numpy(ctx, x, dtype=dt)
        yt = snn.Tensor(ctx, *plan.out_shape(), dtype=dt)
        for _ in range(5):
            plan.run(xt, yt)
        ctx.sync()
        t = snn.Timer(ctx)
        t.start()
        for _ in range(args.reps):
            plan.run(xt, yt)
        t.stop()
        ctx.sync()
        us = 1e3 * t.elapsed_ms() / args.reps
        fl, by = plan.cost()
        tf, gbs = fl / us / 1e6, by / us / 1e3
        t_mfma, t_hbm = fl / PEAK_TF / 1e6, by / PEAK_GBS / 1e3  # us
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        frac = max(t_mfma, t_hbm) / us
        rows.append({"layer": name, "us": us, "tflops": tf, "gbps": gbs, "bound": bound, "frac": frac, "kernel": plan.describe()})
        print("%-38s %9.1f us %7.2f TF/s %8.1f GB/s  %4s-bound %5.1f%%  | %s" % (name, us, tf, gbs, bound, 100 * frac, plan.describe()), flush=True)
        t.destroy()
        plan.destroy()
        xt.free()
        yt.free()
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
