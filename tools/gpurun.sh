#!/bin/bash
# Runs HERE (the build container): stamps the commit the snapshot is taken from (the GPU box has no .git), builds every library in-tree and hands
# the action list to tools/gpu.sh on a GPU box.    usage: tools/gpurun.sh <timeout-seconds> <tag> <action> [<action> ...]
set -e
cd "$(dirname "$0")/.."
T=$1; shift
git rev-parse --short HEAD > .git_head
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "tools/gpu.sh $*"
