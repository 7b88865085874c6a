#!/usr/bin/env python
"""param2json.py -- ncnn .param graph -> the reference's SNN JSON model format, with synthetic weights (the counterpart of the
reference's tools/convertTool for the topology; real weights would come from the matching ncnn .bin).

    python tools/param2json.py model.param out.json [--input H W C] [--seed 1]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shadernn_amd import models, param_import  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("param")
ap.add_argument("json")
ap.add_argument("--input", type=int, nargs=3, default=None, metavar=("H", "W", "C"))
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
net = param_import.import_param(a.param, seed=a.seed, input_shape=tuple(a.input) if a.input else None)
h, w = net["input_hw"]
models.write_json(net, w, h, a.json)
print("%s: %d layers, input %dx%dx%d -> %s" % (net["name"], len(net["layers"]), h, w, net["input_channels"], a.json))
