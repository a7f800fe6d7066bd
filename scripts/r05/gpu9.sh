#!/bin/bash
# round 5, call 9: heaviest-first blend dispatch order: parity, A/B on the headline and the live job
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05i; mkdir -p $O
timeout 1500 python -m pytest tests/test_rast_gpu.py tests/test_render_driver_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest_rast.txt; cat $O/pytest_rast.txt
scripts/gpu_ab.sh $O/blend_order_ab.txt 3 raster "GVF_RAST_BLEND_ORDER=1" "GVF_RAST_BLEND_ORDER=0"
scripts/gpu_ab.sh $O/live_order_ab.txt 3 live "GVF_RAST_BLEND_ORDER=1" "GVF_RAST_BLEND_ORDER=0"
