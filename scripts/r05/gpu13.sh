#!/bin/bash
# round 5, call 13: splat-record stores as whole lines (padding written too / quad-transposed)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05m; mkdir -p $O
V=gvfdiffusion_amd/variants
scripts/gpu_ab.sh $O/pre_rec_ab.txt 3 raster "GVF_X=product" "GVF_LIB=$V/libgvf_hip_pre_rec1.so" "GVF_LIB=$V/libgvf_hip_pre_rec2.so"
for rep in 1 2; do for v in "" pre_rec1 pre_rec2; do
  echo -n "live [$v] " >> $O/pre_rec_ab.txt
  L="GVF_X=1"; [ -n "$v" ] && L="GVF_LIB=$V/libgvf_hip_$v.so"
  env $L python bench.py --live-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_sample'], d['value'])" >> $O/pre_rec_ab.txt
done; done
cat $O/pre_rec_ab.txt
