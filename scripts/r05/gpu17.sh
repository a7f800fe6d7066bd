#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== alone -q"; timeout 900 python -m pytest tests/test_inference_script_gpu.py -m gpu -q 2>&1 | tail -5; echo "rc=${PIPESTATUS[0]}"
echo "== after test_dit_gpu tail, -q -s"; timeout 1500 python -m pytest tests/test_dit_gpu.py tests/test_inference_script_gpu.py -m gpu -q -s -k "split3 or weight_change or script or in_flight or speculation" 2>&1 | tail -30; echo "rc=${PIPESTATUS[0]}"
echo "== full -q with shared activation off"; GVF_RAST_SHARED_ACT=0 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4; echo "rc=${PIPESTATUS[0]}"
