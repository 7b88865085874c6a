#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r5_dbg}; mkdir -p "$O"
timeout 200 python tools/debug_widep.py > "$O/dbg.txt" 2>&1; grep "==" "$O/dbg.txt" | cut -c1-200
DBG_N=3 DBG_H=90 DBG_W=200 timeout 200 python tools/debug_widep.py > "$O/dbg2.txt" 2>&1; grep "==" -A6 "$O/dbg2.txt" | cut -c1-200 | head -40
