#!/bin/bash
# The round's evidence in ONE gpurun call, parametrised (replaces the per-round r03/r04/r04c copies):
#   scripts/evidence.sh <tag> [sections]      sections: any of  raster dit vae bench smoke tests   (default: all, in this order)
# Everything lands in gpurun_out/<tag>/ (scratch); copy what is to be judged into profiles/<tag>_*.  The rasteriser PMC summaries are
# stamped with the rasteriser source hash and copied to profiles/ BEFORE the bench line is taken (bench.py reads the newest one).
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
TAG=${1:?tag}; shift
SECTIONS=${*:-raster dit vae bench smoke tests}
O=gpurun_out/$TAG; mkdir -p $O
has() { [[ " $SECTIONS " == *" $1 "* ]]; }
RB="--steps 3 --warmup 1 --no-cpu-baseline --no-dit --streams 1"
if has raster; then
  scripts/gpu_profile.sh $TAG --steps 10 --warmup 2 --no-cpu-baseline --no-dit --streams 1 > $O/profile.log 2>&1
  cp gpurun_out/prof_$TAG/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
  scripts/gpu_pmc.sh ${TAG}_fetch "FETCH_SIZE" $RB > /dev/null 2>&1
  scripts/gpu_pmc.sh ${TAG}_write "WRITE_SIZE" $RB > /dev/null 2>&1
  F=$(ls gpurun_out/pmc_${TAG}_fetch/*counter_collection.csv | head -1); Wf=$(ls gpurun_out/pmc_${TAG}_write/*counter_collection.csv | head -1)
  cp $F $O/pmc_fetch_counter_collection.csv; cp $Wf $O/pmc_write_counter_collection.csv
  python scripts/pmc_summary.py $F $Wf $O/pmc_raster.json 24 > $O/pmc_raster_summary.txt
  scripts/pmc_py.sh ${TAG}_rast bench.py $RB > /dev/null 2>&1
  python scripts/pmc_sq_summary.py $O/pmc_raster_sq_summary.txt $O/pmc_raster_sq.json gpurun_out/pmc_${TAG}_rast/pass1.csv gpurun_out/pmc_${TAG}_rast/pass2.csv gpurun_out/pmc_${TAG}_rast/pass3.csv > /dev/null 2>&1
  cp $O/pmc_raster.json profiles/${TAG}_pmc_raster.json; cp $O/pmc_raster_sq.json profiles/${TAG}_pmc_raster_sq.json
fi
if has dit; then
  export GVF_BENCH_DIT_CFG3=0 GVF_BENCH_DIT_INFLIGHT=0 GVF_BENCH_DIT_OTHER_DTYPE=0 GVF_BENCH_DIT_HOSTILE=0
  for t in fp16 bf16; do
    GVF_DIT_DTYPE=$t scripts/gpu_profile.sh ${TAG}_dit_$t --dit-only > /dev/null 2>&1
    python scripts/dit_breakdown.py gpurun_out/prof_${TAG}_dit_$t/${TAG}_dit_${t}_kernel_trace.csv auto > $O/dit_kernel_breakdown_$t.txt
    cp gpurun_out/prof_${TAG}_dit_$t/${TAG}_dit_${t}_kernel_stats.csv $O/dit_kernel_stats_$t.csv
    rm -f gpurun_out/prof_${TAG}_dit_$t/*kernel_trace.csv
  done
  GVF_BENCH_DIT_NFE=8 GVF_DIT_DTYPE=fp16 scripts/pmc_py.sh ${TAG}_dit8 bench.py --dit-only --no-cpu-baseline > /dev/null 2>&1
  python scripts/pmc_sq_summary.py $O/pmc_dit_sq_summary.txt - gpurun_out/pmc_${TAG}_dit8/pass1.csv gpurun_out/pmc_${TAG}_dit8/pass2.csv gpurun_out/pmc_${TAG}_dit8/pass3.csv > /dev/null 2>&1
  unset GVF_BENCH_DIT_CFG3 GVF_BENCH_DIT_INFLIGHT GVF_BENCH_DIT_OTHER_DTYPE GVF_BENCH_DIT_HOSTILE
fi
if has vae; then
  for t in bf16 fp16; do
    ( cd /tmp && TMPDIR=/tmp GVF_DIT_DTYPE=$t rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/vae_$t -o vae -- python $OLDPWD/scripts/vae_breakdown.py > $OLDPWD/$O/vae_breakdown_$t.txt 2>&1 )
    find $O/vae_$t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/vae_decode_kernel_stats_$t.csv
    rm -rf $O/vae_$t
  done
  python scripts/bench_gemm_vae.py 2>&1 | grep -v amdgpu.ids > $O/gemm_vae_shapes.txt
fi
has bench && python bench.py 2>$O/bench_err.log | tail -1 > $O/bench_line.json
has smoke && python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
has tests && python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.txt
rm -rf gpurun_out/pmc_${TAG}_*/pass*.csv gpurun_out/prof_$TAG/*kernel_trace.csv
ls -la $O
[ -f $O/bench_line.json ] && head -c 900 $O/bench_line.json && echo
[ -f $O/smoke.txt ] && tail -3 $O/smoke.txt
[ -f $O/pytest_gpu.txt ] && cat $O/pytest_gpu.txt
[ -f $O/dit_kernel_breakdown_fp16.txt ] && cat $O/dit_kernel_breakdown_fp16.txt
exit 0
