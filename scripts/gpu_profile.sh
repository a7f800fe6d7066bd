#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace + stats of one bench.py run; copies the
# summaries into gpurun_out/prof_<tag>/ (scratch) -- commit the ones you want judged under profiles/.
# usage: scripts/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $REPO/bench.py "$@" > "$OUT/bench_under_rocprof.log" 2>&1 < /dev/null
echo "rocprofv3 rc=$?"
for f in $(find /tmp/prof_$TAG -name "*stats*.csv" -o -name "*kernel_trace.csv" 2>/dev/null); do cp "$f" "$OUT/"; done
ls -la "$OUT"
tail -n 1 "$OUT/bench_under_rocprof.log" | cut -c1-400
K=$(ls "$OUT"/*kernel_stats.csv 2>/dev/null | head -n 1)
if [ -n "$K" ]; then head -n 16 "$K" | cut -c1-220; fi
