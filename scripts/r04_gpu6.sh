#!/bin/bash
# round 4, GPU session 6: in-flight capture bisection, whole GPU suite, the driver's bench line, two gloo ranks on one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
python - <<'PY'
import sys, os
sys.path.insert(0, '.')
from gvfdiffusion_amd import _build
stamp = open(_build.STAMP_PATH).read().strip() if os.path.exists(_build.STAMP_PATH) else ''
if stamp != _build.source_hash():
    print('library stale on the box: rebuilding'); _build.build(force=True)
PY
for sw in "REPRO_EAGER=1" "REPRO_DET=1" "REPRO_SOLOCOND=1" "REPRO_SAMPLES=4"; do
  env $sw REPRO_ROUNDS=20 timeout 400 python scripts/inflight_capture_repro.py > $O/repro_$sw.txt 2>&1
  echo "$sw: $(tail -1 $O/repro_$sw.txt | cut -c1-300)"
done
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/tests_all.txt 2>&1
tail -5 $O/tests_all.txt
timeout 900 python bench.py 2>$O/bench_err.log | tail -1 > $O/bench_line.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04f/bench_line.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], d['ms_per_step_serial'], 'roof', d['roofline']['frac'])
print('dit', {k: d['dit'].get(k) for k in ('value', 'ms_per_nfe', 'dtype')}, d['dit']['roofline'], d['dit'].get('other_dtype'), d['dit'].get('in_flight'))
print('e2e', d['end_to_end'].get('value'), d['end_to_end'].get('wall_ms'), d['end_to_end'].get('stage_ms'), d['end_to_end'].get('nfe'))
print('live', d.get('live_render'))
PY
GVF_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 --no-dit --no-cpu-baseline 2>$O/bench_gloo2_err.log | tail -1 > $O/bench_gloo2_line.json
head -c 900 $O/bench_gloo2_line.json; echo
