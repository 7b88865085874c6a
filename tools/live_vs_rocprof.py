#!/usr/bin/env python
"""bench.py's live per-kernel leg against rocprofv3 IN THE SAME PROCESS: runs on the GPU box as
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python bench.py --config cN --also none --detail-out <dir>/detail.json ...
(tools/gpu.sh action `livecheck:cN`) and then compares, for the dominant kernel function of the line's `roofline`, the average launch duration bench.py measured
with its own launch trace (run right after the timed region) with the durations rocprofv3 recorded for the SAME launches -- the last launches of that kernel in
the trace -- and with rocprofv3's --stats average over all launches (pre-heat and warm-up included).   python tools/live_vs_rocprof.py <dir>"""
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    det = json.load(open(os.path.join(d, "detail.json")))
    rf = det["roofline"]
    fn, launches_per_step, steps = rf["kernel"], rf["launches_per_step"], det["traced_steps"]
    n_live = int(round(launches_per_step * steps))
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if fn in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    rows.sort()
    dur = [x for _, x in rows]
    assert len(dur) >= n_live, (len(dur), n_live)
    same = dur[-n_live:]  # the launch-by-launch passes of the live trace are the process's last launches of the kernel
    out = {"config": det["config"]["config_id"], "kernel": fn, "launches_in_rocprof_trace": len(dur), "launches_of_the_live_trace": n_live,
           "live_avg_launch_us": rf["avg_launch_us"], "rocprof_same_launches_avg_us": sum(same) / len(same), "rocprof_same_launches_min_us": min(same),
           "rocprof_same_launches_max_us": max(same), "rocprof_stats_avg_all_launches_us": sum(dur) / len(dur), "rocprof_median_all_launches_us": sorted(dur)[len(dur) // 2],
           "live_over_rocprof_same_launches": rf["avg_launch_us"] / (sum(same) / len(same)), "roofline_frac_on_the_line": rf["frac"], "ms_per_step_under_rocprof": det["ms_per_step"]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
