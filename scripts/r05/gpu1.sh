#!/bin/bash
# round 5, call 1: persistent attn_xt (ubench A/B vs the round-4 binary), the new parity tests, DiT step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05a; mkdir -p $O
{
echo "== round-4 kernel (xt_base.bin)"; scripts/ubench/xt_base.bin 20 1
echo "== persistent build, GVF_ATTN_PERSIST=1"; scripts/ubench/xt_p1.bin 20 1
echo "== persistent build, GVF_ATTN_PERSIST=0 (one item per workgroup)"; GVF_ATTN_PERSIST=0 scripts/ubench/xt_p1.bin 20 1
echo "== XT_PERSIST=0 build"; scripts/ubench/xt_p0.bin 20 1
echo "== round-4 kernel again"; scripts/ubench/xt_base.bin 20 1
echo "== persistent build again"; scripts/ubench/xt_p1.bin 20 1
echo "== persistent build, full check"; scripts/ubench/xt_p1.bin 5
echo "== persistent build, stamps"; scripts/ubench/xt_p1tm.bin 10 1
} > $O/xt_ubench.txt 2>&1
tail -40 $O/xt_ubench.txt
timeout 1500 python -m pytest tests/test_dit_fp16_gpu.py tests/test_dit_gpu.py tests/test_rowblock_temporal_gpu.py -m gpu -x -q -s 2>&1 | grep -E "trained-like|passed|failed|Error|error|full DiT \[" | tail -30 > $O/pytest_dit.txt; cat $O/pytest_dit.txt
timeout 900 python -m pytest tests/test_rast_gpu.py -m gpu -x -q -s -k "full_size or live_shape" 2>&1 | grep -E "config2|live shape|passed|failed|Error|assert" | tail -30 > $O/pytest_rast.txt; cat $O/pytest_rast.txt
timeout 1200 python -m pytest tests/test_inference_script_gpu.py tests/test_distributed.py -m gpu -x -q -s -k "script_runs or bench_n2" 2>&1 | grep -E "bench --gpus|passed|failed|Error|assert" | tail -20 > $O/pytest_misc.txt; cat $O/pytest_misc.txt
scripts/gpu_ab.sh $O/dit_persist_ab.txt 2 dit "GVF_ATTN_PERSIST=1" "GVF_ATTN_PERSIST=0"
