#!/bin/bash
# The gpurun calls of round 6, ONE parametrised script (no more one-shot gpuNN.sh files):
#   local:  scripts/r06.sh run <case> [timeout-s]     rebuilds the library, then  gpurun -- "bash scripts/r06.sh <case>"
#   box:    scripts/r06.sh <case>                     the measurement itself; everything lands in gpurun_out/r06_<case>/
# A/B legs go through scripts/gpu_ab.sh, the round's evidence through scripts/evidence.sh; this file only names the combinations that were run.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
if [ "${1:-}" = "run" ]; then
  python -c "from gvfdiffusion_amd import _build; _build.build()" 2>&1 | grep -v packed-fp32
  python -c "import oracle; oracle.build()"
  exec /usr/local/graft/bin/gpurun --timeout ${3:-1500} -- "bash scripts/r06.sh $2"
fi
CASE=${1:?case}; O=gpurun_out/r06_$CASE; mkdir -p $O
bench_summary() { python -c "
import json,sys; d=json.load(open('$1'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'serial', d['ms_per_step_serial'], 'blend frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
print('stages', d['stage_ms_per_step'])
if 'dit' in d: print('dit', d['dit']['ms_per_nfe'], d['dit'].get('parity_vs_fp32_golden'), 'cfg3', d['dit'].get('cfg3'))
if 'end_to_end' in d: print('e2e', d['end_to_end'].get('wall_ms'), d['end_to_end'].get('stage_ms'))
if 'live_render' in d: print('live', d['live_render']['ms_per_sample'])
if 'sharded_sampling' in d: print('sharded', json.dumps(d['sharded_sampling'])[:1500])
"; }
case $CASE in
  first)    # shared activation arithmetic: raster + sampler GPU tests with per-test durations, then the whole bench line (new anchor / parity fields)
    timeout 900 python -m pytest tests/test_rast_gpu.py tests/test_sampler.py tests/test_render_driver_gpu.py tests/test_pipeline_gpu.py -m gpu -q --durations=30 -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -45 $O/pytest.txt
    timeout 900 python bench.py 2>$O/bench_err.log | tail -1 > $O/bench_line.json; bench_summary $O/bench_line.json; tail -5 $O/bench_err.log ;;
  suite)    # the whole GPU suite with durations
    timeout 1500 python -m pytest tests -m gpu -q --durations=40 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -60 $O/pytest.txt ;;
  batch)    # where does a batched forward spend its time?  kernel traces of the DiT leg at B = 1, 3, 8 (one forward covers B samples) + the GPU suite's durations
    export GVF_BENCH_DIT_CFG3=0 GVF_BENCH_DIT_INFLIGHT=0 GVF_BENCH_DIT_OTHER_DTYPE=0 GVF_BENCH_DIT_HOSTILE=0 GVF_BENCH_DIT_NFE=8
    for B in 1 3 8; do
      GVF_BENCH_DIT_BATCH=$B scripts/gpu_profile.sh r06b$B --dit-only > $O/prof_b$B.log 2>&1
      python scripts/dit_breakdown.py gpurun_out/prof_r06b$B/r06b${B}_kernel_trace.csv auto > $O/dit_kernel_breakdown_b$B.txt
      rm -f gpurun_out/prof_r06b$B/*kernel_trace.csv
      echo "== B=$B"; tail -1 gpurun_out/prof_r06b$B/bench_under_rocprof.log | cut -c1-300; cat $O/dit_kernel_breakdown_b$B.txt
    done
    unset GVF_BENCH_DIT_NFE
    scripts/gpu_ab.sh $O/act_arith_ab.txt 3 raster "GVF_LIB=gvfdiffusion_amd/variants/libgvf_hip_actlibm.so" "GVF_X=1"
    timeout 1500 python -m pytest tests -m gpu -q --durations=40 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -60 $O/pytest.txt ;;
  ubench)   # VERDICT r5 items 6 + 9 (micro-benchmarks before any kernel work), item 4's first measurement, and the adaptive chain after the host-side changes
    scripts/ubench/exp2_pk16.bin > $O/ubench_exp2_pk16.txt 2>&1; cat $O/ubench_exp2_pk16.txt
    scripts/ubench/blend_step.bin > $O/ubench_blend_step.txt 2>&1; cat $O/ubench_blend_step.txt
    for w in live bench; do GVF_LIB=gvfdiffusion_amd/variants/libgvf_hip_blendc.so python scripts/blend_consumed.py $w 2>&1 | grep -v amdgpu.ids > $O/blend_consumed_$w.txt; cat $O/blend_consumed_$w.txt; done
    python scripts/adaptive_host_profile.py 2>&1 | grep -v amdgpu.ids | head -40 > $O/adaptive_host_profile.txt; head -12 $O/adaptive_host_profile.txt
    scripts/gpu_ab.sh $O/e2e.txt 3 e2e "GVF_X=1"
    scripts/gpu_ab.sh $O/dit.txt 2 dit "GVF_X=1"
    timeout 900 python -m pytest tests/test_sampler.py tests/test_vae_gpu.py tests/test_sparse_vae_gpu.py tests/test_dit_gpu.py -m gpu -q --durations=12 -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest.txt ;;
  second)   # the fixed blend-step micro-benchmark, the bench line with the three sharded modes, and the tests touched since (durations with the oracle disk cache)
    scripts/ubench/blend_step.bin > $O/ubench_blend_step.txt 2>&1; cat $O/ubench_blend_step.txt
    timeout 900 python bench.py 2>$O/bench_err.log | tail -1 > $O/bench_line.json; bench_summary $O/bench_line.json; tail -3 $O/bench_err.log
    rm -rf /tmp/gvf_oracle_cache
    timeout 1200 python -m pytest tests/test_sampler.py tests/test_dit_gpu.py tests/test_vae_gpu.py tests/test_sparse_vae_gpu.py tests/test_distributed.py -m gpu -q --durations=12 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest.txt ;;
  clocks)   # is the batched forward clock / power limited?  + the sampler test on the device after the tolerance fix
    python scripts/dit_clock_probe.py 5 2>&1 | grep -v amdgpu.ids > $O/dit_clock_probe.txt; cat $O/dit_clock_probe.txt
    timeout 600 python -m pytest tests/test_sampler.py -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt ;;
  streams)  # raster samples in flight 1..4 (the headline's `value`), the three sharded modes with the full-length warm-up, sampler test
    for n in 2 3 4 2 3; do echo -n "streams $n: "; python bench.py --no-dit --no-cpu-baseline --streams $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_serial'])"; done | tee $O/raster_streams.txt
    GVF_BENCH_DIT_CFG3=0 GVF_BENCH_DIT_INFLIGHT=0 GVF_BENCH_DIT_OTHER_DTYPE=0 GVF_BENCH_DIT_HOSTILE=0 timeout 900 python bench.py --steps 5 2>$O/bench_err.log | tail -1 > $O/bench_line.json; bench_summary $O/bench_line.json
    timeout 600 python -m pytest tests/test_sampler.py -m gpu -q -s 2>&1 | tail -6 ;;
  rbfill)   # VERDICT r5 item 2a priced by ablation: the row-block launches with 3 / 2 of a wave's 4 column tiles refilled per k-step = the weight traffic
            # 64-row / 96-row workgroups would leave (timing only: the results are wrong), at batch 3 and batch 1
    for B in 3 1; do GVF_BENCH_DIT_BATCH=$B GVF_BENCH_DIT_NFE=16 scripts/gpu_ab.sh $O/rowblock_fill_b$B.txt 2 dit "GVF_X=full" "GVF_LIB=gvfdiffusion_amd/variants/libgvf_hip_rbw3.so" "GVF_LIB=gvfdiffusion_amd/variants/libgvf_hip_rbw2.so"; done ;;
  u8)       # uint8 frames from the blend's epilogue: parity test, the live job with / without it, and the headline leg (the kernel gained an argument)
    timeout 900 python -m pytest tests/test_rast_gpu.py tests/test_render_driver_gpu.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -30
    [ "${SKIP_LIVE:-0}" = 1 ] || scripts/gpu_ab.sh $O/live_u8_ab.txt 3 live "GVF_RENDER_FUSED_U8=0" "GVF_RENDER_FUSED_U8=1"
    scripts/gpu_ab.sh $O/raster.txt 3 raster "GVF_X=1" "GVF_LIB=gvfdiffusion_amd/variants/libgvf_hip_nou8.so" ;;
  gemmbk)   # the plain GEMM's existing 64-deep k-tile instantiation on the motion VAE's shapes (K = 768 / 3072): in isolation and inside the decode
    for bk in 0 64 0 64; do echo "== GVF_GEMM_BK=$bk"; GVF_GEMM_BK=$bk python scripts/bench_gemm_vae.py 2>&1 | grep -v amdgpu.ids; GVF_GEMM_BK=$bk python scripts/vae_breakdown.py 2>&1 | grep -v amdgpu.ids | tail -2; done | tee $O/gemm_bk.txt ;;
  hd64)     # the DiT with one head of 64 (per-sub-layer path) + the tests touched since the evidence pass
    timeout 900 python -m pytest tests/test_dit_fp16_gpu.py tests/test_dit_gpu.py tests/test_sampler.py tests/test_distributed.py -m gpu -q -x -s 2>&1 | grep -v "^$" | grep "one head of 64\|passed\|failed\|Error\|assert" | tail -20 ;;
  gemm8)    # the eight-wave 256-wide GEMM in the product: parity tests, the VAE's shapes with it off / on / forced, the decode in place, the e2e leg
    timeout 900 python -m pytest tests/test_dit_gpu.py tests/test_vae_gpu.py tests/test_sparse_vae_gpu.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -12
    for v in 0 1 2 0 1; do echo "== GVF_GEMM8=$v"; GVF_GEMM8=$v python scripts/bench_gemm_vae.py 2>&1 | grep -v amdgpu.ids | cut -c1-80; GVF_GEMM8=$v python scripts/vae_breakdown.py 2>&1 | grep -v amdgpu.ids | tail -1; GVF_DIT_DTYPE=fp16 GVF_GEMM8=$v python scripts/vae_breakdown.py 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $O/gemm8_vae.txt
    scripts/gpu_ab.sh $O/e2e_gemm8.txt 2 e2e "GVF_GEMM8=0" "GVF_GEMM8=1" ;;
  *) echo "unknown case $CASE"; exit 2 ;;
esac
