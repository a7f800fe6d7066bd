#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05p; mkdir -p $O
timeout 600 python -m pytest tests/test_rast_gpu.py -m gpu -x -q -k "blend_dispatch or shared_activation" 2>&1 | tail -5
for i in 1 2; do
  timeout 1500 python -X faulthandler -m pytest tests -m gpu -q > $O/pytest_full_$i.txt 2>&1; echo "run $i rc=$?"; tail -4 $O/pytest_full_$i.txt
done
