#!/usr/bin/env python
"""bench_models.py -- whole-model throughput of BASELINE configs 3-5 shapes on one MI355X through per-layer C-ABI plans:
ResNet-18 224x224 batch 32 (config 3), MobileNetV2 224x224 batch 32 (= the per-GPU share of config 4's batch 256 / 8 GPUs) and
the toy style net at 720p batch 1 (the operator set of config 5, fp32).  Synthetic weights of the real topologies (models.py).
Prints images/s, achieved TFLOP/s and GB/s on the algorithmic (per-layer, unfused) accounting, and the slowest layers.

    python tools/bench_models.py [--reps 20] [--model resnet18|mobilenetv2|candy|style|unet|yolov3-tiny] [--batch N] [--json out.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK_TF, PEAK_GBS = 157.3, 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--model", default=None)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--json", default=None)
    ap.add_argument("--tune", action="store_true", help="SNNHIP_CONV_TUNE=1: time a few (block width, split-K) candidates per convolution at plan creation")
    ap.add_argument("--fp16", action="store_true", help="half tensors + fp16 MFMA convolutions (graphs without depthwise / dense / pooling layers)")
    args = ap.parse_args()
    if args.tune:
        os.environ.setdefault("SNNHIP_CONV_TUNE", "1")  # =2 (set by the caller) also logs every candidate, see tools/report_tune.py
    import shadernn_amd as snn
    from shadernn_amd import models

    snn.load_library()
    ctx = snn.Context(0)
    def zoo(name, shape):
        from shadernn_amd import param_import

        fx = json.load(open(os.path.join(ROOT, "shadernn_amd", "data", "zoo_topologies.json")))[name]
        ops = [{"type": o["type"], "name": o["name"], "inputs": o["inputs"], "outputs": o["outputs"], "params": {int(k): v for k, v in o["params"].items()}}
               for o in fx["ops"]]
        return param_import.from_ops(ops, name=name, seed=1, input_shape=shape)

    cases = [("resnet18", lambda: models.resnet18(seed=1), 32, 224, 224), ("mobilenetv2", lambda: models.mobilenetv2(seed=1), 32, 224, 224),
             ("candy", lambda: zoo("candy-9_simplified-opt", (720, 1280, 3)), 1, 720, 1280),   # the reference zoo's graph (config 5 topology), fp32
             ("style", lambda: models.style_net(seed=1, width=32), 1, 720, 1280),
             # the zoo's U-Net (256 x 256 x 1) and YOLOv3-tiny (416 x 416 x 3, two heads) graphs
             ("unet", lambda: zoo("unet", None), 8, 256, 256), ("yolov3-tiny", lambda: zoo("yolov3-tiny", (416, 416, 3)), 8, 416, 416)]
    out = []
    for name, make, batch, h, w in cases:
        if args.model and args.model != name:
            continue
        batch = args.batch or batch
        net = make()
        r = snn.GraphRunner(ctx, net, batch, h, w, dtype=snn.F16 if args.fp16 else snn.F32)
        r.x.upload(np.random.default_rng(1).random(r.in_shape, dtype=np.float32))
        for _ in range(3):
            r.run_device()
        ctx.sync()
        t = snn.Timer(ctx)
        t.start()
        for _ in range(args.reps):
            r.run_device()
        t.stop()
        ctx.sync()
        ms = t.elapsed_ms() / args.reps
        # the same inference as one captured hipGraph (one host call per batch instead of one per layer)
        with snn.Graph.capture(ctx) as g:
            r.run_device()
        for _ in range(3):
            g.launch()
        ctx.sync()
        t.start()
        for _ in range(args.reps):
            g.launch()
        t.stop()
        ctx.sync()
        ms_graph = t.elapsed_ms() / args.reps
        nodes = g.num_nodes()
        g.destroy()
        # per-layer times: one timed loop per plan
        rows, by_type = [], {}
        for si, (plan, ins, o, layer) in enumerate(r.steps):
            t.start()
            for _ in range(5):
                plan.run(ins if len(ins) > 1 else ins[0], o)
            t.stop()
            ctx.sync()
            f, b = r.step_cost(si)
            rows.append((t.elapsed_ms() / 5 * 1e3, layer["name"], f, b, plan.describe()))
            by_type.setdefault(layer["type"], [0.0, 0, 0.0])
            by_type[layer["type"]][0] += rows[-1][0]
            by_type[layer["type"]][1] += 1
            by_type[layer["type"]][2] += b
        fl, by = r.cost()
        ms_eager, ms = ms, min(ms, ms_graph)
        res = {"model": name, "dtype": "f16" if args.fp16 else "f32", "batch": batch, "input": [batch, h, w, 3], "ms_per_batch": ms, "ms_per_batch_eager_launches": ms_eager,
               "ms_per_batch_hipgraph": ms_graph, "graph_nodes": nodes, "images_per_s": batch / ms * 1e3, "gflop_per_image": fl / batch / 1e9,
               "mb_per_image_unfused": by / batch / 1e6, "tflops": fl / ms / 1e9, "gbps_unfused": by / ms / 1e6,
               "roofline_ms": max(fl / PEAK_TF / 1e9, by / PEAK_GBS / 1e6), "layers": len(r.steps)}
        res["frac_of_roofline"] = res["roofline_ms"] / ms
        print("%-12s eager %.2f ms, hipGraph (%d nodes) %.2f ms" % (name, ms_eager, nodes, ms_graph))
        if args.fp16:
            PEAK = 2500.0
            res["roofline_ms"] = max(fl / PEAK / 1e9, by / PEAK_GBS / 1e6)
            res["frac_of_roofline"] = res["roofline_ms"] / ms
        print("%-12s batch %3d: %8.2f ms/batch  %9.1f images/s  %6.2f TFLOP/s  %7.1f GB/s (unfused accounting)  roofline %.2f ms -> %.1f%%  (%d launches)" %
              (name, batch, ms, res["images_per_s"], res["tflops"], res["gbps_unfused"], res["roofline_ms"], 100 * res["frac_of_roofline"], len(r.steps)), flush=True)
        rows.sort(reverse=True)
        tot = sum(x[0] for x in rows)
        for us, lname, f, b, desc in rows[:(len(rows) if os.environ.get('BENCH_ALL') else 8)]:
            print("     %8.1f us %5.1f%%  %-16s %6.2f TF/s %7.1f GB/s | %s" % (us, 100 * us / tot, lname, f / us / 1e6, b / us / 1e3, desc[:110]))
        print("     by layer type: " + "  ".join("%s x%d %.0f us (%.0f GB/s)" % (k, v[1], v[0], v[2] / v[0] / 1e3) for k, v in sorted(by_type.items(), key=lambda kv: -kv[1][0])))
        res["by_type_us"] = {k: v[0] for k, v in by_type.items()}
        res["top_layers"] = [{"us": us, "layer": lname, "kernel": desc} for us, lname, f, b, desc in rows[:8]]
        out.append(res)
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
