#!/bin/bash
# row sums by 16x16x32 selector MFMA (the product) against the 4x4x4 form; attention + DiT parity tests; DiT step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04j; mkdir -p $O
python -c "import gvfdiffusion_amd._build as b; b.build(verbose=False)" >/dev/null 2>&1
for r in 1 2 3; do for v in base s1; do echo "== $v" >> $O/xt_sum.txt; timeout 120 scripts/ubench/xt_$v.bin 30 1 | cut -c1-30,95-200 >> $O/xt_sum.txt; done; done
timeout 120 scripts/ubench/xt_base.bin 5 | cut -c1-150 >> $O/xt_sum.txt
cat $O/xt_sum.txt
timeout 1500 python -m pytest tests/test_dit_gpu.py tests/test_dit_fp16_gpu.py tests/test_rowblock_temporal_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_dit.txt
export GVF_BENCH_DIT_CFG3=0 GVF_BENCH_DIT_INFLIGHT=0
timeout 600 python bench.py --dit-only --no-cpu-baseline 2>>$O/err.log | tail -1 > $O/dit_line.json
python -c "import json; d=json.load(open('$O/dit_line.json')); print({k:d[k] for k in d if k in ('ms_per_nfe','value','other_dtype','dtype')})"
