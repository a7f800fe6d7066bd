#!/bin/bash
# round 5, call 2: item-boundary stamps of the persistent attention, split-bf16 condition GEMMs, ordered fp16 caches, hostile bench legs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05b; mkdir -p $O
{ scripts/ubench/xt_base.bin 20 1; scripts/ubench/xt_p1.bin 20 1; scripts/ubench/xt_base.bin 20 1; scripts/ubench/xt_p1.bin 20 1; scripts/ubench/xt_p1tm.bin 10 1; scripts/ubench/xt_p1.bin 5; } > $O/xt_stamps.txt 2>&1; cat $O/xt_stamps.txt
timeout 1500 python -m pytest tests/test_dit_fp16_gpu.py tests/test_dit_gpu.py tests/test_rowblock_temporal_gpu.py -m gpu -x -q -s 2>&1 | grep -E "trained-like|split3|key order|passed|failed|Error|error|full DiT \[|assert" | tail -40 > $O/pytest_dit.txt; cat $O/pytest_dit.txt
python scripts/bench_prepare_conditions.py 2>&1 | grep -v amdgpu.ids > $O/prepare_conditions.txt; cat $O/prepare_conditions.txt
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prep_prof -o prep -- python $OLDPWD/scripts/bench_prepare_conditions.py > /dev/null 2>&1 ); find /tmp/prep_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/prepare_conditions_kernel_stats.csv; head -12 $O/prepare_conditions_kernel_stats.csv | cut -c1-160
GVF_BENCH_DIT_CFG3=0 GVF_BENCH_DIT_INFLIGHT=0 GVF_BENCH_DIT_OTHER_DTYPE=1 GVF_BENCH_DIT_HOSTILE=1 python bench.py --dit-only --no-cpu-baseline 2>$O/dit_err.log | tail -1 > $O/dit_line.json
python -c "
import json; d=json.load(open('$O/dit_line.json')); print('dit', d['ms_per_nfe'], d['dtype'], d['softmax_guard']); print('other', d['other_dtype']); print('hostile', json.dumps(d['trained_like_weights'], indent=1))"
python bench.py --e2e-only 2>>$O/dit_err.log | tail -1 > $O/e2e_line.json; cat $O/e2e_line.json
