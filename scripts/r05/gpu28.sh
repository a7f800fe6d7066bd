#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05t; mkdir -p $O
timeout 900 python -m pytest tests/test_dit_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q -k "step_size_test or adaptive or pipeline or chain" 2>&1 | tail -3
scripts/gpu_ab.sh $O/e2e_fused_norm_ab.txt 3 e2e "GVF_DPM_FUSED_NORM=1" "GVF_DPM_FUSED_NORM=0"
