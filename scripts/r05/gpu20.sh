#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05t; mkdir -p $O
scripts/gpu_ab.sh $O/e2e_hold_ab.txt 3 e2e "GVF_DIT_HOLD_PVER=1" "GVF_DIT_HOLD_PVER=0"
timeout 900 python -m pytest tests/test_dit_gpu.py tests/test_inference_script_gpu.py -m gpu -x -q 2>&1 | tail -3
