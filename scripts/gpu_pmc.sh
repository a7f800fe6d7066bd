#!/bin/bash
# usage: scripts/gpu_pmc.sh <tag> "<counters>" <bench args...>   (counters in their own pass; kernel-trace only)
set -u
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
timeout 900 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmc_$TAG -o $TAG -- python $REPO/bench.py "$@" > "$OUT/log.txt" 2>&1 < /dev/null
echo "rocprofv3 rc=$?"
for f in $(find /tmp/pmc_$TAG -name "*counter_collection.csv" 2>/dev/null); do cp "$f" "$OUT/"; done
ls -la "$OUT"
