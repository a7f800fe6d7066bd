#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05n; mkdir -p $O
timeout 1500 python -m pytest tests/test_rast_gpu.py tests/test_render_driver_gpu.py tests/test_pipeline_gpu.py tests/test_rast_bwd_gpu.py tests/test_inference_script_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest_rast.txt; cat $O/pytest_rast.txt
for i in 1 2 3; do python bench.py --live-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('live', d['ms_per_sample'], d['value'])"; done
