#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05y; mkdir -p $O
for w in live bench; do GVF_LIB=gvfdiffusion_amd/variants/libgvf_hip_sortstats.so python scripts/sort_stats.py $w 2>&1 | grep -v amdgpu.ids >> $O/sort_stats.txt; done; cat $O/sort_stats.txt
