#!/usr/bin/env python
"""Copies what a tools/gpu.sh <tag> tests prof:c2 prof:c3 prof:c4 prof:c5 bench run left under gpurun_out/ into profiles/ (tracked): the rocprofv3 summaries of the c2 / c3 / c4 / c5 bench commands, the
merged per-kernel PMC traffic file bench.py reads (profiles/pmc_latest.json, keyed to the kernel-source fingerprint) and one bench line per config.

    python tools/collect_profiles.py r02z [--round r02]
"""
import argparse
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--round", default="r02")
    a = ap.parse_args()
    out, prof = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    names = {"c1": "%s_c1_conv3x3_rocprofv3_summary.md", "c2": "%s_c2_rocprofv3_summary.md", "c3": "%s_c3_resnet18_rocprofv3_summary.md", "c4": "%s_c4_mobilenetv2_rocprofv3_summary.md",
             "c5": "%s_c5_candy_fp16_rocprofv3_summary.md"}
    pmc = {}
    for c, fmt in names.items():
        d = os.path.join(out, "%s_%s" % (a.tag, c))
        if os.path.exists(os.path.join(d, "summary.md")):
            shutil.copy(os.path.join(d, "summary.md"), os.path.join(prof, fmt % a.round))
        if os.path.exists(os.path.join(d, "pmc_by_kernel.json")):
            for k, v in json.load(open(os.path.join(d, "pmc_by_kernel.json"))).items():
                v["bench_config"] = c
                pmc.setdefault(k, v)
    if pmc:
        json.dump(pmc, open(os.path.join(prof, "pmc_latest.json"), "w"), indent=1)
    for c in ("c1", "c2", "c3", "c4", "c5", "all"):
        src = os.path.join(out, a.tag, "bench_%s.json" % c)
        if os.path.exists(src) and os.path.getsize(src):
            shutil.copy(src, os.path.join(prof, "%s_bench_%s.json" % (a.round, c)))
    print("pmc entries:", len(pmc), "fingerprints:", sorted({v.get("csrc_sha16") for v in pmc.values()}))


if __name__ == "__main__":
    main()
