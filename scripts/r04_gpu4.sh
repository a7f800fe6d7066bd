#!/bin/bash
# round 4, GPU session 4: full DiT / raster parity after the temporal rework, prefetch A/B, in-flight capture reproducer, f32-MFMA overlap ubench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
# safety net: rebuild on the box if the shipped library does not match the shipped sources (an edit between `gpurun` and the snapshot)
python - <<'PY'
import sys
sys.path.insert(0, '.')
from gvfdiffusion_amd import _build
import os
stamp = open(_build.STAMP_PATH).read().strip() if os.path.exists(_build.STAMP_PATH) else ''
if stamp != _build.source_hash():
    print('library stale on the box: rebuilding'); _build.build(force=True)
PY
true
true
timeout 1500 python -m pytest tests/test_rowblock_temporal_gpu.py tests/test_dit_gpu.py tests/test_dit_fp16_gpu.py -x -q > $O/tests_dit.txt 2>&1
tail -4 $O/tests_dit.txt
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -s -k "full_size" > $O/pipeline_full.txt 2>&1
grep "configs\[3\]\|passed\|failed" $O/pipeline_full.txt
export GVF_BENCH_DIT_CFG3=0 GVF_BENCH_DIT_INFLIGHT=0 GVF_BENCH_DIT_OTHER_DTYPE=0
for pf in 1 0 1 0; do GVF_DIT_PREFETCH=$pf timeout 600 python bench.py --dit-only --no-cpu-baseline 2>>$O/bench_dit_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prefetch $pf', d.get('ms_per_nfe'), d.get('value'), d.get('roofline',{}).get('frac'))" >> $O/bench_dit_prefetch.txt; done
cat $O/bench_dit_prefetch.txt
REPRO_ROUNDS=30 timeout 600 python scripts/inflight_capture_repro.py > $O/repro_default.txt 2>&1
tail -4 $O/repro_default.txt | cut -c1-500
REPRO_ROUNDS=30 REPRO_NO_EMPTY=1 REPRO_NO_GC=1 timeout 600 python scripts/inflight_capture_repro.py > $O/repro_noempty_nogc.txt 2>&1
tail -3 $O/repro_noempty_nogc.txt | cut -c1-500
REPRO_ROUNDS=30 REPRO_PAUSE=1 timeout 600 python scripts/inflight_capture_repro.py > $O/repro_pause.txt 2>&1
tail -3 $O/repro_pause.txt | cut -c1-500
timeout 300 python bench.py --live-only > $O/live_render.json 2>$O/live_err.log; cat $O/live_render.json
