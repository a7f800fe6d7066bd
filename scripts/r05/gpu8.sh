#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05h; mkdir -p $O
GVF_LIB=gvfdiffusion_amd/variants/libgvf_hip_blendt.so python scripts/blend_stamps.py 2>&1 | grep -v amdgpu.ids > $O/blend_stamps.txt; cat $O/blend_stamps.txt
