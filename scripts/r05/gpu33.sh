#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05ae; mkdir -p $O
timeout 900 python -m pytest tests/test_tile_sort_gpu.py tests/test_rast_gpu.py -m gpu -x -q 2>&1 | tail -3
scripts/gpu_ab.sh $O/sort_prefetch_ab.txt 3 live "GVF_X=prefetch" "GVF_LIB=gvfdiffusion_amd/variants/libgvf_hip_nopf.so"
