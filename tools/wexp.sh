cd "$GRAFT_REPO_ROOT"
L="--fp16 --only adhoc --reps 30 --shape 16,183,323,128,128,3,1"
for v in "" w1b w1b_a1 w1b_a2 w1b_a4 w1b_a7 w1b_a15 w2b_a15; do
  if [ -z "$v" ]; then unset SNNHIP_LIB_PATH; else export SNNHIP_LIB_PATH=$GRAFT_REPO_ROOT/build/abl/libsnnhip_$v.so; fi
  echo "== [${v:-product}] $(python tools/bench_layers.py $L 2>/dev/null | grep adhoc | cut -c1-95)"
done
