#!/bin/bash
# round 4, GPU session 5: blend forms A/B (classic | matrix fp16-split | matrix f32), preprocess frames-per-workgroup sweep, DiT kernel breakdown
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04e; mkdir -p $O
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
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['ms_per_step_serial'], d['stage_ms_per_step'])"; }
for rep in 1 2; do
  GVF_RAST_BLEND=1 timeout 300 python bench.py --no-dit --no-cpu-baseline 2>>$O/err.log | tail -1 | line "classic" >> $O/blend_ab.txt
  GVF_RAST_BLEND=2 timeout 300 python bench.py --no-dit --no-cpu-baseline 2>>$O/err.log | tail -1 | line "matrix-h16" >> $O/blend_ab.txt
  GVF_RAST_BLEND=2 GVF_RAST_MX_F32=1 timeout 300 python bench.py --no-dit --no-cpu-baseline 2>>$O/err.log | tail -1 | line "matrix-f32" >> $O/blend_ab.txt
done
cat $O/blend_ab.txt
timeout 900 python -m pytest tests/test_rast_gpu.py tests/test_render_driver_gpu.py -x -q 2>&1 | tail -5 > $O/tests_rast.txt; cat $O/tests_rast.txt
B=gvfdiffusion_amd/csrc/build
for fb in 8 12 24 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -DGVF_PRE_FB=$fb -c gvfdiffusion_amd/csrc/rast.hip -o $B/rast.o 2>>$O/err.log
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gvfdiffusion_amd/libgvf_hip.so $B/*.o 2>>$O/err.log
  GVF_RAST_BLEND=1 timeout 300 python bench.py --no-dit --no-cpu-baseline 2>>$O/err.log | tail -1 | line "PRE_FB=$fb" >> $O/pre_fb.txt
done
cat $O/pre_fb.txt
export GVF_BENCH_DIT_CFG3=0 GVF_BENCH_DIT_INFLIGHT=0 GVF_BENCH_DIT_OTHER_DTYPE=0
GVF_DIT_DTYPE=fp16 scripts/gpu_profile.sh dit_fp16 --dit-only > /dev/null 2>&1
python scripts/dit_breakdown.py gpurun_out/prof_dit_fp16/dit_fp16_kernel_trace.csv 36 > $O/dit_kernel_breakdown_fp16.txt
rm -f gpurun_out/prof_dit_fp16/*kernel_trace.csv
cat $O/dit_kernel_breakdown_fp16.txt
