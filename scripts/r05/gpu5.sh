#!/bin/bash
# round 5, call 5: preprocess XCD-aware order / delta prefetch / frames per workgroup A/B; blend per-phase stamps (per-wave rows)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05e; mkdir -p $O
V=gvfdiffusion_amd/variants
timeout 1200 python -m pytest tests/test_rast_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -4 > $O/pytest_rast.txt; cat $O/pytest_rast.txt
GVF_LIB=$V/libgvf_hip_blendt.so python scripts/blend_stamps.py 2>&1 | grep -v amdgpu.ids > $O/blend_stamps.txt; cat $O/blend_stamps.txt
scripts/gpu_ab.sh $O/pre_ab.txt 2 raster "GVF_X=xcd1" "GVF_LIB=$V/libgvf_hip_prexcd0.so" "GVF_LIB=$V/libgvf_hip_prepf.so" "GVF_LIB=$V/libgvf_hip_prefb2.so" "GVF_LIB=$V/libgvf_hip_prefb8.so" "GVF_LIB=$V/libgvf_hip_prefb2pf.so"
