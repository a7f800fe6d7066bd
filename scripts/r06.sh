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
  *) echo "unknown case $CASE"; exit 2 ;;
esac
