#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05x; mkdir -p $O
( cd /tmp && TMPDIR=/tmp GVF_LIVE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/live_prof -o live -- python $OLDPWD/bench.py --live-only > /dev/null 2>&1 ); find /tmp/live_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/live_render_kernel_stats_streams1.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05x/live_render_kernel_stats_streams1.csv')))
for r in rows[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'], "%.1f ms per job"%(int(r['TotalDurationNs'])/1e6/2), "avg %.1f us"%(float(r['AverageNs'])/1e3))
print("sum %.1f ms per job" % (sum(int(r['TotalDurationNs']) for r in rows)/2e6))
PY
