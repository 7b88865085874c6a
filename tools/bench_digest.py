#!/usr/bin/env python
"""One line per bench.py JSON line found under a directory (or in the files named): value, ms/step, roofline fractions, parity, wait semantics."""
import glob
import json
import os
import sys


def _one(tag, d):
    r, p = d.get("roofline") or {}, d.get("parity") or {}
    t = d.get("timing") or {}
    s = "%s: %.1f img/s %.4f ms/step (min %.4f max %.4f, R=%s) | whole-step roofline %.3f, launches %.3f | dominant %s [%s] %.3f share %.2f traffic ratio %s | parity ok=%s abs %.2e" % (
        tag, d.get("value", 0), d.get("ms_per_step", 0), t.get("min_ms_per_step", 0), t.get("max_ms_per_step", 0), t.get("repeats"), d.get("frac_of_whole_step_roofline", 0),
        d.get("frac_of_sum_of_launch_rooflines", 0), (r.get("kernel") or "")[:48], r.get("bound"), r.get("frac", 0), r.get("share_of_gpu_time", 0),
        ("%.3f" % r["traffic_over_algorithmic_bytes"]) if r.get("traffic_over_algorithmic_bytes") else r.get("traffic"), p.get("ok"), p.get("max_abs_err", float("nan")))
    for k in (d.get("kernels") or [])[:8]:
        s += "\n    %8.1f us/step x%-5.1f %5.1f%% %-5s %s  %s" % (k["us_per_step"], k["launches_per_step"], 100 * k["share_of_gpu_time"], k["bound"],
                                                                ("%.3f" % k["frac"]) if k.get("frac") is not None else "  -  ", k["function"])
    return s


def digest(path):
    try:
        line = open(path).read().strip().split("\n")[-1]
        full = path[:-5] + ".detail.json"  # since round 6 the printed line is a summary (< 8 KB); the full record sits beside it
        d = json.load(open(full)) if os.path.exists(full) else json.loads(line)
        d["_line_bytes"] = len(line)
    except Exception as e:
        return "%s: unreadable (%s)" % (path, e)
    cb, w = d.get("cpu_baseline") or {}, d.get("wait_semantics") or {}
    s = _one("%s %s [%s]" % (os.path.basename(path), d["config"].get("config_id"), d.get("value_mode")), d)
    s += "\n    cpu %.2f img/s @%s threads | printed line %d bytes" % (cb.get("value", 0), cb.get("cores"), d["_line_bytes"])
    if w:
        s += "\n    wait: in flight %.4f ms, sync/inference %.4f ms (polling) %.4f ms (blocking)" % (
            w["inflight"]["ms_per_step"], w["sync_per_inference"]["ms_per_step"], w["sync_per_inference_blocking_wait"]["ms_per_step"])
    for c, rec in (d.get("configs") or {}).items():
        s += "\n  " + (_one("configs." + c, rec) if "error" not in rec else "configs.%s: ERROR %s" % (c, rec["error"]))
    if d.get("configs_note"):
        s += "\n  " + d["configs_note"][-60:]
    return s


def main():
    for a in sys.argv[1:]:
        files = sorted(f for f in glob.glob(os.path.join(a, "bench*.json")) if not f.endswith(".detail.json")) if os.path.isdir(a) else [a]
        for f in files:
            if os.path.getsize(f):
                print(digest(f))
            else:
                print("%s: empty" % f)


if __name__ == "__main__":
    main()
