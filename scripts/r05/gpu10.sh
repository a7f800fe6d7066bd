#!/bin/bash
# round 5, call 10: the live render job, one chunk at a time under rocprofv3 (clean kernel durations), and its wall time with 1 / 2 chunks in flight
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05j; mkdir -p $O
( cd /tmp && TMPDIR=/tmp GVF_LIVE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/live_prof -o live -- python $OLDPWD/bench.py --live-only > /dev/null 2>&1 ); find /tmp/live_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/live_render_kernel_stats_streams1.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05j/live_render_kernel_stats_streams1.csv')))
for r in rows[:16]:
    print(r['Name'][:70].ljust(70), r['Calls'], "%.1f ms per job"%(int(r['TotalDurationNs'])/1e6/2), "avg %.1f us"%(float(r['AverageNs'])/1e3))
print("sum %.1f ms per job" % (sum(int(r['TotalDurationNs']) for r in rows)/2e6))
PY
for s in 1 2 3; do echo -n "streams=$s "; GVF_LIVE_STREAMS=$s python bench.py --live-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_sample'], d['value'])"; done
python - <<'PY'
import torch, os, sys
sys.path.insert(0, os.getcwd())
from gvfdiffusion_amd import rasterizer as R
import bench
PY
