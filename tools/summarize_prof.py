#!/usr/bin/env python
"""Summarises rocprofv3 CSV output (kernel stats + PMC passes) of tools/profile_gpu.sh into markdown + JSON."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def rows(pattern):
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def short(name):
    n = name.replace("void snnhip::(anonymous namespace)::", "").replace("snnhip::(anonymous namespace)::", "")
    n = n.replace("(anonymous namespace)::", "")
    n = n.split("(")[0]
    return n[:110]


def main():
    out = sys.argv[1]
    res = {"kernels": {}}
    print("# rocprofv3 summary (%s)\n" % os.path.basename(out))
    print("## kernel stats (--kernel-trace --stats)\n")
    print("| kernel | calls | avg us | min us | max us | total ms | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows(out + "/trace/**/*kernel_stats.csv"):
        name = short(r["Name"])
        avg = float(r["AverageNs"]) / 1e3
        print("| %s | %s | %.2f | %.2f | %.2f | %.3f | %s |" % (name, r["Calls"], avg, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                            float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
        res["kernels"].setdefault(name, {})["avg_us"] = avg
        res["kernels"][name]["calls"] = int(r["Calls"])
    # per-dispatch durations of the same trace: the --stats average above includes the pre-heat and warm-up launches (clocks still ramping: the max column);
    # the timed region of bench.py is the tail of the run, so the median and the mean of the second half of a kernel's launches are what its `roofline` leg
    # (measured after the timed region, clocks up) has to agree with
    per = defaultdict(list)
    for r in rows(out + "/trace/**/*kernel_trace.csv"):
        per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    if per:
        print("\n## steady state (same trace, per-dispatch durations in launch order)\n")
        print("| kernel | calls | median us | mean of the 2nd half us | mean of the 1st half us |")
        print("|---|---|---|---|---|")
        for name, lst in sorted(per.items(), key=lambda kv: -sum(d for _, d in kv[1])):
            lst.sort()
            d = [x for _, x in lst]
            h = len(d) // 2
            med = sorted(d)[len(d) // 2]
            second = sum(d[h:]) / max(1, len(d) - h)
            first = sum(d[:h]) / max(1, h) if h else second
            print("| %s | %d | %.2f | %.2f | %.2f |" % (name, len(d), med, second, first))
            res["kernels"].setdefault(name, {}).update({"median_us": med, "steady_mean_us": second})
    # PMC: average per dispatch per kernel
    for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(lambda: defaultdict(int))
        for r in rows(out + "/%s/**/*counter_collection.csv" % sub):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
        if not acc:
            continue
        print("\n## %s (average per dispatch)\n" % sub)
        for k in acc:
            print("* **%s**" % k)
            for c in sorted(acc[k]):
                v = acc[k][c] / max(cnt[k][c], 1)
                print("  * %s = %.6g" % (c, v))
                res["kernels"].setdefault(k, {})[c] = v
    # derived
    print("\n## derived\n")
    for k, d in res["kernels"].items():
        line = []
        if "GRBM_GUI_ACTIVE" in d and "avg_us" in d:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            d["eff_clock_ghz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / (d["avg_us"] * 1e3)
            line.append("eff_clock_GHz=%.3f" % d["eff_clock_ghz"])
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["SQ_VALU_MFMA_BUSY_CYCLES"]:
            # busy cycles summed over 1024 SIMDs / elapsed cycles
            d["mfma_pipe_util"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (d["GRBM_GUI_ACTIVE"] / 8.0)
            line.append("mfma_pipe_util=%.3f" % d["mfma_pipe_util"])
        if "SQ_WAVE_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
            d["avg_waves_per_simd"] = d["SQ_WAVE_CYCLES"] * 4.0 / 1024.0 / (d["GRBM_GUI_ACTIVE"] / 8.0)  # SQ_WAVE_CYCLES counts quad-cycles
            line.append("avg_waves_per_simd=%.2f" % d["avg_waves_per_simd"])
        if "FETCH_SIZE" in d:
            # rocprofv3 reports KiB; gfx950 wide coalesced reads are tallied at half their size (MI355X_MICROARCH.md HBM)
            d["hbm_read_bytes_raw"] = d["FETCH_SIZE"] * 1024
            d["hbm_read_bytes_corrected_x2"] = d["FETCH_SIZE"] * 1024 * 2
            line.append("FETCH raw=%.1f MB (x2 corrected=%.1f MB)" % (d["hbm_read_bytes_raw"] / 1e6, d["hbm_read_bytes_corrected_x2"] / 1e6))
        if "WRITE_SIZE" in d:
            d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
            line.append("WRITE=%.1f MB" % (d["hbm_write_bytes"] / 1e6))
        if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE"):
            line.append("lds_conflict_frac=%.3f" % (d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]))
        if line:
            print("* **%s**: %s" % (k, "; ".join(line)))
    json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
    # compact per-kernel HBM traffic, keyed by the bare kernel function name (template arguments stripped): bench.py reads
    # profiles/pmc_latest.json in this format for the `roofline.traffic` field
    pmc = {}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from shadernn_amd import fingerprint

    sha = fingerprint.csrc_sha16()  # bench.py only reports `traffic` from a file taken with the kernel sources it runs
    head = os.environ.get("SNN_GIT_HEAD", "")
    for k, d in res["kernels"].items():
        if "hbm_read_bytes_corrected_x2" in d and "hbm_write_bytes" in d:
            base = k.split("<")[0].strip()
            rec = {"hbm_bytes_per_launch": d["hbm_read_bytes_corrected_x2"] + d["hbm_write_bytes"],
                         "read_bytes_x2_corrected": d["hbm_read_bytes_corrected_x2"], "write_bytes": d["hbm_write_bytes"],
                         "csrc_sha16": sha, "git_head": head, "avg_us_kernel_trace": d.get("avg_us"), "mfma_pipe_util": d.get("mfma_pipe_util"), "eff_clock_ghz": d.get("eff_clock_ghz"),
                         "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_gpu.sh); FETCH_SIZE KiB x2 "
                                   "per MI355X_MICROARCH.md HBM section"}
            # one record per INSTANTIATION (`name<args>`, blanks removed: the key a plan description's `kernel=` tag can name exactly), and the bare
            # function name for the first one seen (kernels with a single instantiation in the profiled command)
            pmc[k.replace(" ", "")] = rec
            pmc.setdefault(base, rec)
    json.dump(pmc, open(os.path.join(out, "pmc_by_kernel.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
