#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05u; mkdir -p $O
for rep in 1 2; do for s in 1 2 3 4; do echo -n "streams=$s " >> $O/raster_streams.txt; python bench.py --no-dit --no-cpu-baseline --streams $s 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_serial'], d['value'])" >> $O/raster_streams.txt; done; done; cat $O/raster_streams.txt
