#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05af; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_full.txt 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest_full.txt
python bench.py 2>/dev/null | tail -1 > $O/bench_line.json; python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['dit']['ms_per_nfe'], d['end_to_end']['wall_ms'], d['live_render']['ms_per_sample'], d['cpu_baseline']['value'])"
