#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c; mkdir -p $O
B=gvfdiffusion_amd/csrc/build
for v in "" "-DRB_DBG_VREG" "-DRB_DBG_KEEP_RS" "-DRB_DBG_VREG -DRB_DBG_KEEP_RS"; do
  echo "=== variant [$v]" >> $O/dbg.txt
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $v -c gvfdiffusion_amd/csrc/rowblock.hip -o $B/rowblock.o 2>>$O/dbg.txt
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gvfdiffusion_amd/libgvf_hip.so $B/*.o 2>>$O/dbg.txt
  timeout 300 python -m pytest tests/test_rowblock_temporal_gpu.py -q -s 2>&1 | grep "temporal section\|passed\|failed" | head -40 >> $O/dbg.txt
done
cat $O/dbg.txt | tail -80
timeout 1200 python -m pytest tests/test_rast_gpu.py -x -q 2>&1 | tail -15 > $O/tests_rast.txt
cat $O/tests_rast.txt
for b in 1 2 1 2; do GVF_RAST_BLEND=$b timeout 300 python bench.py --no-dit --no-cpu-baseline 2>>$O/bench_rast_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('blend_algo $b', d['value'], d['ms_per_step'], d['ms_per_step_serial'], d['stage_ms_per_step'])" >> $O/bench_rast_blend.txt; done
cat $O/bench_rast_blend.txt
