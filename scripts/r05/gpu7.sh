#!/bin/bash
# round 5, call 7: blend with the colour accumulation on the matrix pipe (v_mfma_f32_4x4x1): parity through the whole rasteriser suite, then A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05g; mkdir -p $O
V=gvfdiffusion_amd/variants
GVF_LIB=$V/libgvf_hip_blendmfma.so timeout 1500 python -m pytest tests/test_rast_gpu.py tests/test_render_driver_gpu.py tests/test_pipeline_gpu.py tests/test_rast_bwd_gpu.py -m gpu -x -q 2>&1 | tail -12 > $O/pytest_rast_mfma.txt; cat $O/pytest_rast_mfma.txt
scripts/gpu_ab.sh $O/blend_mfma_ab.txt 3 raster "GVF_X=product" "GVF_LIB=$V/libgvf_hip_blendmfma.so"
scripts/gpu_ab.sh $O/live_mfma_ab.txt 2 live "GVF_X=product" "GVF_LIB=$V/libgvf_hip_blendmfma.so"
