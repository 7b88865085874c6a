#!/bin/bash
# irb_image_kernel block census (round 6): wall-clock start / staging-done / end (100 MHz ticks) and XCC / SE / CU of every block of a few launches.
#   tools/exp_one.sh irb_fused.hip census:-DSNNHIP_IRBI_TRACE=2;  tools/gpu.sh <tag> sh:r6_icensus.sh
cd "$GRAFT_REPO_ROOT"
export SNNHIP_LIB_PATH="$GRAFT_REPO_ROOT/build/abl/libsnnhip_census.so"
for b in b11 b07; do
  python tools/bench_irb.py --batch 256 --fused-only --reps 2 --only $b 2>/dev/null | grep '^irbc' > gpurun_out/$1/census_$b.txt
  python - "$1" "$b" <<'PY'
import sys, collections
tag, b = sys.argv[1], sys.argv[2]
rows = [l.split() for l in open("gpurun_out/%s/census_%s.txt" % (tag, b))]
recs = [(int(r[1]), int(r[3]), int(r[5]), int(r[7]), int(r[9]), int(r[10]), int(r[11])) for r in rows]
# launches: 256 consecutive blocks per launch in time order
recs.sort(key=lambda r: r[4])
n = len(recs) // 256
for L in range(n):
    g = recs[L * 256:(L + 1) * 256]
    t0 = min(r[4] for r in g); t1 = max(r[6] for r in g)
    starts = sorted(r[4] - t0 for r in g); ends = sorted(r[6] - t0 for r in g); life = sorted(r[6] - r[4] for r in g); stage = sorted(r[5] - r[4] for r in g)
    q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
    print("%s launch %d: span %.2f us | block start min/med/p90/max %.2f %.2f %.2f %.2f | end min/med/max %.2f %.2f %.2f | life min/med/max %.2f %.2f %.2f | staging med/max %.2f %.2f | CUs used %d" % (
        b, L, (t1 - t0) / 100.0, starts[0] / 100.0, q(starts, .5) / 100.0, q(starts, .9) / 100.0, starts[-1] / 100.0, ends[0] / 100.0, q(ends, .5) / 100.0, ends[-1] / 100.0,
        life[0] / 100.0, q(life, .5) / 100.0, life[-1] / 100.0, q(stage, .5) / 100.0, stage[-1] / 100.0, len({(r[1], r[2], r[3]) for r in g})))
    byx = collections.defaultdict(list)
    for r in g: byx[r[1]].append((r[6] - r[4]) / 100.0)
    print("   life by XCC (median us):", " ".join("%d:%.1f" % (x, sorted(v)[len(v) // 2]) for x, v in sorted(byx.items())))
PY
done
