#!/usr/bin/env python3
"""Summarise `SNNHIP_CONV_TUNE=2` logs (stderr of any run): per convolution geometry, the heuristic configuration and its time against the
fastest candidate.  Usage: SNNHIP_CONV_TUNE=2 python tools/bench_models.py --model resnet18 --tune 2>&1 >/dev/null | python tools/report_tune.py"""
import re
import sys
from collections import OrderedDict

geo = OrderedDict()
for line in sys.stdin:
    m = re.match(r"\[snnhip tune\] (.*?) \| bn=(\d+) splitK=(\d+) -> (.*) : ([-\d.]+) us( \(heuristic\))?", line)
    if not m:
        continue
    g, bn, sk, desc, us, heur = m.groups()
    t = re.search(r"x (\d+)oc .* splitK=(\d+)", desc)
    cfg = "bn%s/k%s" % (t.group(1), t.group(2))
    d = geo.setdefault(g, {"c": []})
    d["c"].append((float(us), cfg))
    if heur:
        d["h"] = (float(us), cfg)
tot_h = tot_b = 0.0
for g, d in geo.items():
    best = min(d["c"])
    h = d.get("h", best)
    tot_h += h[0]
    tot_b += best[0]
    flag = "" if best[0] > 0.95 * h[0] else "  <-- %.0f%%" % (100 * (1 - best[0] / h[0]))
    print("%-44s heur %-9s %7.1f us | best %-9s %7.1f us%s   all: %s" % (g, h[1], h[0], best[1], best[0], flag, " ".join("%s=%.0f" % (c, u) for u, c in sorted(d["c"], key=lambda x: x[1]))))
print("sum over distinct geometries: heuristic %.1f us, best %.1f us" % (tot_h, tot_b))
