#!/usr/bin/env python
"""Writes tests/golden/graph_stages.json: the stage graph (order, names, execution type, output dims, input references) that the host mirror builds for the
five BASELINE configs (`bench.make_net` -> models.write_json -> dp::loadFromJsonModel -> dp::generateInferenceGraph, read back through
snn_graph_summary).  The stage order names dump files and timers, so it is part of the contract with the reference (core/src/ic2/dp.cpp:389-640); the
fixture was generated with the round-4 dp.cpp and pins the round-5 rewrite of that file to the same graphs.   python tests/golden/make_graph_stages.py"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from shadernn_amd import host, models


def stage_graphs():
    out = {}
    tmp = tempfile.mkdtemp()
    for cfg in ("c1", "c2", "c3", "c4", "c5"):
        c = bench.CONFIGS[cfg]
        h, w = c["hw"]
        path = models.write_json(bench.make_net(cfg), w, h, os.path.join(tmp, cfg + ".json"), bin_weights=True)
        rows = host.graph_summary(path, w, h, c["cin"])
        out[cfg] = [[r["index"], r["name"].split("/")[-1], r["loc"], list(r["dims"]), r["inputs"]] for r in rows]
    return out


if __name__ == "__main__":
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_stages.json"), "w") as f:
        json.dump(stage_graphs(), f, separators=(",", ":"))
    print("written")
