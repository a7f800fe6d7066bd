set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
# 1. default bench line
python bench.py 2>gpurun_out/r04/bench_err.log | tail -1 > gpurun_out/r04/bench_line.json
# 2. raster kernel stats (serial, one stream)
scripts/gpu_profile.sh r04 --steps 10 --warmup 2 --no-cpu-baseline --no-dit --streams 1 > gpurun_out/r04/profile.log 2>&1
cp gpurun_out/prof_r04/*kernel_stats.csv gpurun_out/r04/bench_kernel_stats.csv 2>/dev/null
# 3. PMC traffic
scripts/gpu_pmc.sh fetch5 "FETCH_SIZE" --steps 3 --warmup 1 --no-cpu-baseline --no-dit --streams 1 > /dev/null 2>&1
scripts/gpu_pmc.sh write5 "WRITE_SIZE" --steps 3 --warmup 1 --no-cpu-baseline --no-dit --streams 1 > /dev/null 2>&1
F=$(ls gpurun_out/pmc_fetch5/*counter_collection.csv | head -1); Wf=$(ls gpurun_out/pmc_write5/*counter_collection.csv | head -1)
cp $F gpurun_out/r04/pmc_fetch_counter_collection.csv; cp $Wf gpurun_out/r04/pmc_write_counter_collection.csv
python scripts/pmc_summary.py $F $Wf gpurun_out/r04/pmc_raster.json 24 > gpurun_out/r04/pmc_raster_summary.txt
# 4. SQ counters raster
scripts/pmc_py.sh rast5 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dit --streams 1 > /dev/null 2>&1
python scripts/pmc_sq_summary.py gpurun_out/r04/pmc_raster_sq_summary.txt gpurun_out/r04/pmc_raster_sq.json gpurun_out/pmc_rast5/pass1.csv gpurun_out/pmc_rast5/pass2.csv gpurun_out/pmc_rast5/pass3.csv > /dev/null 2>&1
# 5. DiT breakdown per dtype
export GVF_BENCH_DIT_CFG3=0 GVF_BENCH_DIT_INFLIGHT=0 GVF_BENCH_DIT_OTHER_DTYPE=0
for t in fp16 bf16; do
  GVF_DIT_DTYPE=$t scripts/gpu_profile.sh dit_$t --dit-only > /dev/null 2>&1
  python scripts/dit_breakdown.py gpurun_out/prof_dit_$t/dit_${t}_kernel_trace.csv auto > gpurun_out/r04/dit_kernel_breakdown_$t.txt
  cp gpurun_out/prof_dit_$t/dit_${t}_kernel_stats.csv gpurun_out/r04/dit_kernel_stats_$t.csv
  rm -f gpurun_out/prof_dit_$t/*kernel_trace.csv
done
# 6. SQ counters DiT (fp16)
GVF_BENCH_DIT_NFE=8 GVF_DIT_DTYPE=fp16 scripts/pmc_py.sh dit8 bench.py --dit-only --no-cpu-baseline > /dev/null 2>&1
python scripts/pmc_sq_summary.py gpurun_out/r04/pmc_dit_sq_summary.txt - gpurun_out/pmc_dit8/pass1.csv gpurun_out/pmc_dit8/pass2.csv gpurun_out/pmc_dit8/pass3.csv > /dev/null 2>&1
# 7. motion-VAE decode by kernel, both operand types; the decoder cross attention alone (attn_xt64 against attn.hip's K/V-resident kernel); GEMM yardstick
for t in bf16 fp16; do
  ( cd /tmp && TMPDIR=/tmp GVF_DIT_DTYPE=$t rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/vae_$t -o vae -- python $GRAFT_REPO_ROOT/scripts/vae_breakdown.py > $GRAFT_REPO_ROOT/gpurun_out/r04/vae_breakdown_$t.txt 2>&1 )
  find gpurun_out/r04/vae_$t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04/vae_decode_kernel_stats_$t.csv
  rm -rf gpurun_out/r04/vae_$t
done
{ for d in 0 1; do scripts/ubench/x64_vf.bin 5 $d; scripts/ubench/kvres_base.bin 5 $d; done; for L in 64 256; do scripts/ubench/x64_vf.bin 5 0 43648 $L; done
  scripts/ubench/x64_noq.bin 5 0; scripts/ubench/x64_noqst.bin 5 0; scripts/ubench/x64_vf.bin 2 0 8192 512 40 0; scripts/ubench/x64_vf.bin 2 1 333 40 1 0; } > gpurun_out/r04/attn_xt64_ubench.txt 2>&1
python scripts/bench_gemm_vae.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/gemm_vae_shapes.txt
rm -rf gpurun_out/pmc_*/pass*.csv gpurun_out/prof_r04/*kernel_trace.csv
ls -la gpurun_out/r04
head -c 600 gpurun_out/r04/bench_line.json; echo
cat gpurun_out/r04/pmc_raster_summary.txt | head -20
cat gpurun_out/r04/dit_kernel_breakdown_fp16.txt
