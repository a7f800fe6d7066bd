#!/bin/bash
# round 5, call 11: re-take the evidence of the first session of the round (its gpurun_out was lost with the container): persistent attention ubench + item-boundary
# stamps, prepare_conditions timings and kernel list, the DiT parity tests on the trained-like weights
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05k; mkdir -p $O
{ echo "== XT_PERSIST=0 build (one item per workgroup)"; scripts/ubench/xt_p0.bin 20 1; echo "== persistent build"; scripts/ubench/xt_p1.bin 20 1; echo "== XT_PERSIST=0 build"; scripts/ubench/xt_p0.bin 20 1; echo "== persistent build"; scripts/ubench/xt_p1.bin 20 1; echo "== persistent build, stamps (-DXT_TIMING)"; scripts/ubench/xt_p1tm.bin 10 1; echo "== persistent build, full check"; scripts/ubench/xt_p1.bin 5 | tail -4; } > $O/xt_persist_ubench.txt 2>&1; cat $O/xt_persist_ubench.txt
python scripts/bench_prepare_conditions.py 2>&1 | grep -v amdgpu.ids > $O/prepare_conditions.txt; cat $O/prepare_conditions.txt
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prep_prof -o prep -- python $OLDPWD/scripts/bench_prepare_conditions.py > /dev/null 2>&1 ); find /tmp/prep_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/prepare_conditions_kernel_stats.csv; head -14 $O/prepare_conditions_kernel_stats.csv | cut -c1-150
timeout 1500 python -m pytest tests/test_dit_fp16_gpu.py tests/test_dit_gpu.py -m gpu -q -s -k "trained_like or hostile or key_order or split3 or guard" 2>&1 | grep -E "trained-like|split3|key order|passed|failed|Error|full DiT \[|FAILED|guard" | tail -30 > $O/pytest_dit_hostile.txt; cat $O/pytest_dit_hostile.txt
