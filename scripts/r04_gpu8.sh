#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
python - <<'PY'
import sys, os
sys.path.insert(0, '.')
from gvfdiffusion_amd import _build
stamp = open(_build.STAMP_PATH).read().strip() if os.path.exists(_build.STAMP_PATH) else ''
if stamp != _build.source_hash():
    print('library stale on the box: rebuilding'); _build.build(force=True)
PY
mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04/tests_all.txt 2>&1
tail -4 gpurun_out/r04/tests_all.txt
bash scripts/r04_evidence.sh > gpurun_out/r04/evidence.log 2>&1
tail -40 gpurun_out/r04/evidence.log
