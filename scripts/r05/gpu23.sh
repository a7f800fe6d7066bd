#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05w; mkdir -p $O
export GVF_BENCH_CAP_MULT=2.6
scripts/gpu_ab.sh $O/headline_two_stage_ab.txt 3 raster "GVF_RAST_SHARED_ALWAYS=0" "GVF_RAST_SHARED_ALWAYS=1" "GVF_RAST_SHARED_ALWAYS=1 GVF_RAST_SLOT_ORDER=0"
