#!/bin/bash
# round 4, GPU session 2: fixed solo-wave ubench, MLP-variant regression bisect, parity of the reworked temporal section, weight prefetch A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
scripts/ubench/solo_wave.bin > $O/solo_wave.txt 2>&1
for b in at_8f8e692 at_28522f0 at_acf34ac base; do for c in 1 3; do echo "== $b case $c" >> $O/rb_bisect.txt; timeout 120 scripts/ubench/rb_$b.bin 50 $c >> $O/rb_bisect.txt 2>&1; done; done
timeout 1500 python -m pytest tests/test_rowblock_temporal_gpu.py tests/test_dit_gpu.py tests/test_dit_fp16_gpu.py tests/test_rast_gpu.py -x -q > $O/tests_dit.txt 2>&1
tail -5 $O/tests_dit.txt
export GVF_BENCH_DIT_CFG3=0 GVF_BENCH_DIT_INFLIGHT=0 GVF_BENCH_DIT_OTHER_DTYPE=0
for pf in 1 0 1 0; do GVF_DIT_PREFETCH=$pf timeout 600 python bench.py --dit-only --no-cpu-baseline 2>>$O/bench_dit_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prefetch $pf', d.get('ms_per_nfe'), d.get('value'), d.get('roofline',{}).get('frac'))" >> $O/bench_dit_prefetch.txt; done
cat $O/bench_dit_prefetch.txt
GVF_DIT_DTYPE=fp16 scripts/gpu_profile.sh dit_fp16 --dit-only > /dev/null 2>&1
python scripts/dit_breakdown.py gpurun_out/prof_dit_fp16/dit_fp16_kernel_trace.csv 36 > $O/dit_kernel_breakdown_fp16.txt
rm -f gpurun_out/prof_dit_fp16/*kernel_trace.csv
cat $O/dit_kernel_breakdown_fp16.txt
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -s -k "full_size" > $O/pipeline_full.txt 2>&1
grep "configs\[3\]" $O/pipeline_full.txt
REPRO_ROUNDS=40 timeout 900 python scripts/inflight_capture_repro.py > $O/repro_default.txt 2>&1
tail -3 $O/repro_default.txt
