#!/bin/bash
# after attn_xt64 + fold + 16x16x32 row sums + head_dim-64 swizzle: full GPU suite, decode in both operand types, end-to-end leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04n; mkdir -p $O
python -c "import gvfdiffusion_amd._build as b; b.build(verbose=False)" >/dev/null 2>&1
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 600 python bench.py --e2e-only --no-cpu-baseline 2>>$O/err.log | tail -1 > $O/e2e_line.json
python -c "import json; d=json.load(open('$O/e2e_line.json')); e=d.get('end_to_end', d); print({k: e.get(k) for k in ('wall_ms','nfe','stage_ms','value')})"
