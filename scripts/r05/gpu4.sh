#!/bin/bash
# round 5, call 4: (a) row-block refill-fraction ablation (upper bound of a column split across two CUs), (b) blend per-phase stamps,
# (c) persistent attention A/B on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05d; mkdir -p $O
{
for v in base whalf wq now base whalf; do echo "== rb_$v"; scripts/ubench/rb_$v.bin 20 2>&1 | grep -E "us  |OK|FAIL" ; done
} > $O/rb_refill_fraction.txt 2>&1
cat $O/rb_refill_fraction.txt
GVF_LIB=gvfdiffusion_amd/variants/libgvf_hip_blendt.so python scripts/blend_stamps.py 2>&1 | grep -v amdgpu.ids > $O/blend_stamps.txt; cat $O/blend_stamps.txt
scripts/gpu_ab.sh $O/dit_persist_ab.txt 3 dit "GVF_ATTN_PERSIST=1" "GVF_ATTN_PERSIST=0"
