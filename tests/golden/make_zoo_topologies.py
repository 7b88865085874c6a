#!/usr/bin/env python
"""Regenerates shadernn_amd/data/zoo_topologies.json from the reference's model zoo (run in the build container, where
/root/reference exists): the ncnn `.param` text of the BASELINE model graphs plus U-Net and YOLOv3-tiny parsed into op lists (type, name, blobs,
numeric parameters).  Data only -- the weights are Git-LFS pointers in the reference and are not needed for the topology."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from shadernn_amd import param_import  # noqa: E402

ZOO = "/root/reference/modelzoo"
FILES = {"resnet18_cifar10": "Resnet18/resnet18_cifar10.param", "mobilenetV2": "MobileNetV2/mobilenetV2.param",
         "candy-9_simplified-opt": "StyleTransfer/candy-9_simplified-opt.param", "unet": "U-Net/unet.param",
         "yolov3-tiny": "Yolov3-tiny/yolov3-tiny.param"}

out = {}
for name, rel in FILES.items():
    ops = param_import.parse_param(open(os.path.join(ZOO, rel)).read())
    out[name] = {"source": "modelzoo/" + rel, "ops": [{"type": o["type"], "name": o["name"], "inputs": o["inputs"], "outputs": o["outputs"],
                                                       "params": {str(k): v for k, v in o["params"].items()}} for o in ops]}
json.dump(out, open(os.path.join(ROOT, "shadernn_amd", "data", "zoo_topologies.json"), "w"), separators=(",", ":"))
print({k: len(v["ops"]) for k, v in out.items()})
