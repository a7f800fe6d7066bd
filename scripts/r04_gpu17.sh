#!/bin/bash
# preprocess HOIST variant: parity + A/B (live job, forced off / auto) + headline forced on / auto
# needs the dropped variant: git apply scripts/patches/r04_pre_hoist.patch (then rebuild); the product does not carry it
mkdir -p gpurun_out/r04live
python -m pytest tests/test_rast_gpu.py tests/test_render_driver_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -3
GVF_PRE_HOIST=1 python -m pytest tests/test_rast_gpu.py tests/test_render_driver_gpu.py -m gpu -x -q 2>&1 | tail -2
{ for i in 1 2 3; do
  for m in 0 auto; do
    if [ $m = auto ]; then unset GVF_PRE_HOIST; else export GVF_PRE_HOIST=$m; fi
    echo -n "live hoist=$m  "; python bench.py --live-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_sample'])"
  done
done
for m in auto 1; do
  if [ $m = auto ]; then unset GVF_PRE_HOIST; else export GVF_PRE_HOIST=$m; fi
  echo -n "headline hoist=$m  "; python bench.py --no-dit --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step_serial'], d['stage_ms_per_step']['preprocess'])"
done; } | tee gpurun_out/r04live/pre_hoist_ab.txt
