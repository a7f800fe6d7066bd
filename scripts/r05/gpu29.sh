#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05aa; mkdir -p $O
for rep in 1 2; do for pad in 0 6144 10240 16384; do for st in 2 3; do echo -n "pad=$pad streams=$st " >> $O/blend_pad.txt; GVF_BLEND_LDS_PAD=$pad python bench.py --no-dit --no-cpu-baseline --streams $st --steps 60 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_serial'], d['stage_ms_per_step']['blend'], d['value'])" >> $O/blend_pad.txt; done; done; done; cat $O/blend_pad.txt
