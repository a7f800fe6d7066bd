#!/bin/bash
# tile sort: the large class's lower half as its own launch (MODE 3): parity + A/B of the live render job on one box
mkdir -p gpurun_out/r04live
python -m pytest tests/test_tile_sort_gpu.py tests/test_rast_gpu.py tests/test_render_driver_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  for m in 0 1; do
    echo -n "medium=$m  "; GVF_TILE_SORT_MEDIUM=$m python bench.py --live-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_sample'])"
  done
done | tee gpurun_out/r04live/tile_sort_medium_ab.txt
cd /tmp; export TMPDIR=/tmp
for m in 0 1; do
GVF_TILE_SORT_MEDIUM=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m$m -o live -- python $GRAFT_REPO_ROOT/bench.py --live-only > /dev/null 2>&1
cp /tmp/prof_m$m/live_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r04live/live_kernel_stats_medium$m.csv
grep -E "tile_sort|blend_kernel" /tmp/prof_m$m/live_kernel_stats.csv | cut -d, -f1-4 | sed 's/(HIP_vector.*)"/"/' | cut -c1-120
done
