#!/usr/bin/env python
"""bench_irb.py -- MobileNetV2's inverted-residual blocks (BASELINE configs[3]) one at a time: the fused kernel (irb_fused.hip, chain rule G)
against the three / four separate layers, at --batch images.   python tools/bench_irb.py [--batch 64] [--only b02] [--reps 20]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name, H, C, Ch, Co, stride, residual
BLOCKS = [("b01", 112, 16, 96, 24, 2, False), ("b02", 56, 24, 144, 24, 1, True), ("b03", 56, 24, 144, 32, 2, False), ("b04", 28, 32, 192, 32, 1, True),
          ("b06", 28, 32, 192, 64, 2, False), ("b07", 14, 64, 384, 64, 1, True), ("b10", 14, 64, 384, 96, 1, False), ("b11", 14, 96, 576, 96, 1, True),
          ("b13", 14, 96, 576, 160, 2, False), ("b14", 7, 160, 960, 160, 1, True), ("b16", 7, 160, 960, 320, 1, False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--fused-only", action="store_true")
    args = ap.parse_args()
    import shadernn_amd as snn

    snn.load_library()
    ctx = snn.Context(0)
    rng = np.random.default_rng(1)
    N = args.batch

    def timeit(fn):
        for _ in range(3):
            fn()
        ctx.sync()
        t = snn.Timer(ctx)
        t.start()
        for _ in range(args.reps):
            fn()
        t.stop()
        ctx.sync()
        us = 1e3 * t.elapsed_ms() / args.reps
        t.destroy()
        return us

    for name, H, C, Ch, Co, s, res in BLOCKS:
        if args.only and args.only != name:
            continue
        r = lambda *sh, sc=1.0: (rng.standard_normal(sh) * sc).astype(np.float32)
        bn = lambda c: {"beta": r(c, sc=0.1), "gamma": (1 + r(c, sc=0.1)), "mean": r(c, sc=0.1), "var": (1 + np.abs(r(c, sc=0.1))).astype(np.float32)}
        pe = snn.conv2d_plan(ctx, N, H, H, r(Ch, C, 1, 1, sc=C ** -0.5), r(Ch, sc=0.1), act="relu6", bn=bn(Ch))
        pd = snn.conv2d_plan(ctx, N, H, H, r(Ch, 3, 3, sc=1 / 3), r(Ch, sc=0.1), stride=s, pads=(1, 1, 1, 1), act="relu6", bn=bn(Ch), depthwise=True)
        _, OH, OW, _ = pd.out_shape()
        pp = snn.conv2d_plan(ctx, N, OH, OW, r(Co, Ch, 1, 1, sc=Ch ** -0.5), r(Co, sc=0.1), bn=bn(Co))
        x = snn.Tensor.from_numpy(ctx, rng.random((N, H, H, C), dtype=np.float32))
        th, td, tp, ty = (snn.Tensor(ctx, N, H, H, Ch), snn.Tensor(ctx, N, OH, OW, Ch), snn.Tensor(ctx, N, OH, OW, Co), snn.Tensor(ctx, N, OH, OW, Co))
        if res:
            pa = snn.add_plan(ctx, N, OH, OW, Co)
            fused = snn.graph_fuse(ctx, [(pe, [-1], False), (pd, [0], False), (pp, [1], False), (pa, [2, -1], True)])[3][0]
        else:
            pa = None
            try:
                fused = snn.chain_plan(ctx, [pe, pd, pp])
            except snn.SnnHipError:
                fused = None

        def separate():
            pe.run(x, th)
            pd.run(th, td)
            pp.run(td, tp)
            if pa:
                pa.run([tp, x], ty)

        fl = pe.cost()[0] + pd.cost()[0] + pp.cost()[0]
        t_sep = None if args.fused_only else timeit(separate)
        t_f = timeit(lambda: fused.run(x, ty)) if fused is not None and "irb_fused" in fused.describe() else None
        d = fused.describe() if t_f else ""
        tile = d.split("tile=")[1].split(" ")[0] if "tile=" in d else ("band " + d.split("band per block (")[1].split(")")[0]) if "band per block" in d else "image" if "image per block" in d else "-"
        print("%s %dx%d %d->%d->%d s%d%s b%d: separate %s us, fused[%s per wave] %s us (%.1f TF/s)" % (
            name, H, H, C, Ch, Co, s, " +add" if res else "", N, "%.1f" % t_sep if t_sep else "-", tile, "%.1f" % t_f if t_f else "n/a",
            fl / t_f / 1e6 if t_f else 0.0), flush=True)
        for t in (x, th, td, tp, ty):
            t.free()


if __name__ == "__main__":
    main()
