#!/usr/bin/env python
"""One line per bench.py JSON line found under a directory (or in the files named): value, ms/step, roofline fractions, parity, wait semantics."""
import glob
import json
import os
import sys


def digest(path):
    try:
        d = json.loads(open(path).read().strip().split("\n")[-1])
    except Exception as e:
        return "%s: unreadable (%s)" % (path, e)
    r, cb, p, w = d.get("roofline") or {}, d.get("cpu_baseline") or {}, d.get("parity") or {}, d.get("wait_semantics") or {}
    s = "%s %s: %.1f img/s %.4f ms/step [%s] | whole-step roofline %.3f, launches %.3f | dominant %s %s %.3f traffic %s | parity ok=%s abs %.2e | cpu %.2f @%s" % (
        os.path.basename(path), d["config"].get("config_id"), d["value"], d["ms_per_step"], d.get("value_mode"), d.get("frac_of_whole_step_roofline", 0),
        d.get("frac_of_sum_of_launch_rooflines", 0), (r.get("kernel") or "")[:48], r.get("bound"), r.get("frac", 0), r.get("traffic"), p.get("ok"),
        p.get("max_abs_err", float("nan")), cb.get("value", 0), cb.get("cores"))
    if w:
        s += "\n    wait: in flight %.4f ms, sync/inference %.4f ms (polling) %.4f ms (blocking)" % (
            w["inflight"]["ms_per_step"], w["sync_per_inference"]["ms_per_step"], w["sync_per_inference_blocking_wait"]["ms_per_step"])
    for k in (d.get("kernels") or [])[:8]:
        s += "\n    %8.1f us x%-3d %s" % (k["avg_us"], k["launches"], k["kernel"][:150])
    return s


def main():
    for a in sys.argv[1:]:
        files = sorted(glob.glob(os.path.join(a, "bench*.json"))) if os.path.isdir(a) else [a]
        for f in files:
            if os.path.getsize(f):
                print(digest(f))
            else:
                print("%s: empty" % f)


if __name__ == "__main__":
    main()
