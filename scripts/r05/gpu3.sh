#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c; mkdir -p $O
{ scripts/ubench/xt_base.bin 20 1; scripts/ubench/xt_p1.bin 20 1; scripts/ubench/xt_base.bin 20 1; scripts/ubench/xt_p1.bin 20 1; scripts/ubench/xt_p1tm.bin 10 1; scripts/ubench/xt_p1.bin 5 | tail -3; } > $O/xt_stamps.txt 2>&1; cat $O/xt_stamps.txt
timeout 1500 python -m pytest tests/test_dit_fp16_gpu.py tests/test_dit_gpu.py tests/test_rowblock_temporal_gpu.py -m gpu -q -s 2>&1 | grep -E "trained-like|split3|key order|passed|failed|Error|error|full DiT \[|assert|FAILED" | tail -40 > $O/pytest_dit.txt; cat $O/pytest_dit.txt
python scripts/bench_prepare_conditions.py 2>&1 | grep -v amdgpu.ids > $O/prepare_conditions.txt; cat $O/prepare_conditions.txt
scripts/gpu_ab.sh $O/dit_persist_ab.txt 2 dit "GVF_ATTN_PERSIST=1" "GVF_ATTN_PERSIST=0"
