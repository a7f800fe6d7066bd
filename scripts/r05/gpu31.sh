#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05ac; mkdir -p $O
for rep in 1 2 3; do for lp in 0 1; do for st in 2 3; do echo -n "lowprio=$lp streams=$st " >> $O/blend_lowprio.txt; GVF_RAST_BLEND_LOWPRIO=$lp python bench.py --no-dit --no-cpu-baseline --streams $st --steps 60 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_serial'], d['value'])" >> $O/blend_lowprio.txt; done; done; done; cat $O/blend_lowprio.txt
for lp in 0 1 0 1; do echo -n "live lowprio=$lp "; GVF_RAST_BLEND_LOWPRIO=$lp python bench.py --live-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_sample'], d['value'])"; done | tee -a $O/blend_lowprio.txt
