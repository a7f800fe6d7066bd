#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04i; mkdir -p $O
export GVF_BENCH_DIT_CFG3=0 GVF_BENCH_DIT_INFLIGHT=0 GVF_BENCH_DIT_OTHER_DTYPE=0
for rep in 1 2 3; do for mt in 1 0; do GVF_DIT_MODTABLE=$mt timeout 600 python bench.py --dit-only --no-cpu-baseline 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('modtable $mt', d.get('ms_per_nfe'), d.get('value'))" >> $O/modtable_ab.txt; done; done
cat $O/modtable_ab.txt
