#!/bin/bash
# round 5: row-segment sweeps of the marching kernels at Candy's micro-batch-16 shapes (the planners' static fill model against measurement)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r5_segs}; mkdir -p "$O"
timeout 600 python tools/bench_upconv.py --stats --reps 10 --rounds 3 --segs 0,1,2,3,4,5,6,7,8,10,12,16 > "$O/upconv_segs.txt" 2>&1; cat "$O/upconv_segs.txt" | cut -c1-120
for v in auto 2 3 4 6 8 12; do
  for k in SNNHIP_ROWFOLD_SEGS SNNHIP_S2MARCH_SEGS; do
    if [ "$v" = auto ] && [ "$k" = SNNHIP_S2MARCH_SEGS ]; then continue; fi
    spec="$k=$v"; [ "$v" = auto ] && spec="SNNHIP_UNUSED=1"
    env $spec timeout 600 python bench.py --config c5 --also none --no-cpu-baseline --layer-table 0 --repeats 1 > "$O/bench_c5_${k}_$v.json" 2> "$O/bench_c5_${k}_$v.err" || tail -3 "$O/bench_c5_${k}_$v.err"
    echo "== $spec"; python tools/bench_digest.py "$O/bench_c5_${k}_$v.json" | grep 'img/s\|s2march\|rowfold'
  done
done
