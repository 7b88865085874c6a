#!/bin/bash
# round 5: census of the persistent wide kernel's blocks (experiment build -DSNNHIP_WIDEP_TRACE, a 15-image layer is its marker): XCD, CU, first / last tick (100 MHz), tiles
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r5_census}; mkdir -p "$O"
export SNNHIP_LIB_PATH=$GRAFT_REPO_ROOT/build/abl/libsnnhip_wptrace.so
timeout 300 python tools/bench_layers.py --fp16 --only adhoc --shape 15,187,327,128,128,3,1 --reps 1 > "$O/census.txt" 2>&1
grep -c wpblk "$O/census.txt"
python - "$O/census.txt" <<'PY'
import sys, re, collections
rows = []
for l in open(sys.argv[1]):
    m = re.match(r'wpblk (\d+) xcc (\d+) hw ([0-9a-f]+) t0 (\d+) t1 (\d+) tiles (\d+)', l)
    if m: rows.append((int(m.group(1)), int(m.group(2)), int(m.group(3), 16), int(m.group(4)), int(m.group(5)), int(m.group(6))))
# several launches print: keep the last launch (largest t0 cluster)
rows.sort(key=lambda r: r[3])
if not rows: sys.exit(0)
# split launches by gaps in t0 > 2000 ticks
launches, cur = [], [rows[0]]
for r in rows[1:]:
    if r[3] - cur[-1][3] > 2000: launches.append(cur); cur = []
    cur.append(r)
launches.append(cur)
L = launches[-1]
t00 = min(r[3] for r in L); tend = max(r[4] for r in L)
print("launches", len(launches), "blocks in last", len(L), "duration ticks(10ns)", tend - t00)
bycu = collections.defaultdict(list)
for b, xcc, hw, t0, t1, tiles in L:
    cu = (xcc, (hw >> 8) & 0xf, (hw >> 12) & 0xf)  # (xcc, cu id bits, sh/se bits) -- whatever identifies the CU
    bycu[(xcc, hw >> 8)].append((b, t0 - t00, t1 - t00, tiles))
first_end, second_end, solo = [], [], []
for cu, bl in bycu.items():
    bl.sort(key=lambda x: x[1])
    ends = sorted(x[2] for x in bl)
    if len(bl) == 2:
        first_end.append(ends[0]); second_end.append(ends[1]); solo.append(ends[1] - ends[0])
print("CUs", len(bycu), "with two blocks", len(first_end))
import statistics as st
if first_end:
    print("first block of a CU ends  : mean %.0f  min %d max %d" % (st.mean(first_end), min(first_end), max(first_end)))
    print("second block of a CU ends : mean %.0f  min %d max %d" % (st.mean(second_end), min(second_end), max(second_end)))
    print("one block alone on its CU : mean %.0f ticks (%.1f %% of the launch)" % (st.mean(solo), 100.0 * st.mean(solo) / (tend - t00)))
older = [(b, t1 - t00, tiles) for b, xcc, hw, t0, t1, tiles in L if b < 256]
younger = [(b, t1 - t00, tiles) for b, xcc, hw, t0, t1, tiles in L if b >= 256]
for name, g in (("blocks 0..255", older), ("blocks 256..", younger)):
    for tl in sorted(set(x[2] for x in g)):
        e = [x[1] for x in g if x[2] == tl]
        print("%s with %d tiles: %d blocks, end mean %.0f min %d max %d" % (name, tl, len(e), st.mean(e), min(e), max(e)))
PY
