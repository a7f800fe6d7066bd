#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05ab; mkdir -p $O
for rep in 1 2; do for cfg in "96 2" "128 2" "192 2" "64 3" "96 3" "128 3" "64 4" "256 2"; do set -- $cfg; echo -n "chunk=$1 streams=$2 " >> $O/live_sweep.txt; GVF_LIVE_CHUNK=$1 GVF_LIVE_STREAMS=$2 python bench.py --live-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_sample'], d['value'])" >> $O/live_sweep.txt; done; done; cat $O/live_sweep.txt
