#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05s; mkdir -p $O
scripts/gpu_ab.sh $O/blend_order_tail_ab.txt 3 raster "GVF_RAST_BLEND_ORDER_TAIL=0" "GVF_RAST_BLEND_ORDER_TAIL=1" "GVF_RAST_BLEND_ORDER_TAIL=2" "GVF_RAST_BLEND_ORDER_TAIL=4" "GVF_RAST_BLEND_ORDER_TAIL=24"
python scripts/adaptive_host_profile.py 2>&1 | grep -v amdgpu.ids | head -45 > $O/adaptive_host_profile.txt; cat $O/adaptive_host_profile.txt
