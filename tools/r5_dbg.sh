#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r5_dbg}; mkdir -p "$O"
timeout 200 python tools/debug_widep.py > "$O/dbg.txt" 2>&1; tail -60 "$O/dbg.txt" | cut -c1-230
for t in strict syncepi; do echo "#### $t"; SNNHIP_LIB_PATH=$GRAFT_REPO_ROOT/build/abl/libsnnhip_$t.so timeout 200 python tools/debug_widep.py 2>&1 | grep "==" | cut -c1-200; done
