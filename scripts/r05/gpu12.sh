#!/bin/bash
# round 5, call 12: what do preprocess_kernel's stores cost?  (timing-only ablations: results are wrong)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05l; mkdir -p $O
V=gvfdiffusion_amd/variants
scripts/gpu_ab.sh $O/pre_store_ablation.txt 2 raster "GVF_X=product" "GVF_LIB=$V/libgvf_hip_pre_norec.so" "GVF_LIB=$V/libgvf_hip_pre_nobin.so" "GVF_LIB=$V/libgvf_hip_pre_nostore.so" "GVF_LIB=$V/libgvf_hip_pre_binlin.so"
for v in "" pre_norec pre_nobin pre_nostore pre_binlin; do
  echo -n "live streams=1 [$v] " >> $O/pre_store_ablation.txt
  L=""; [ -n "$v" ] && L="GVF_LIB=$V/libgvf_hip_$v.so"
  env $L GVF_LIVE_STREAMS=1 python bench.py --live-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_sample'], d['value'])" >> $O/pre_store_ablation.txt
done
cat $O/pre_store_ablation.txt
