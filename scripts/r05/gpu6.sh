#!/bin/bash
# round 5, call 6: shared activation (parity tests, live-job A/B, live kernel stats), branchy blend step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05f; mkdir -p $O
V=gvfdiffusion_amd/variants
timeout 1500 python -m pytest tests/test_rast_gpu.py tests/test_render_driver_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest_rast.txt; cat $O/pytest_rast.txt
scripts/gpu_ab.sh $O/live_shared_ab.txt 3 live "GVF_RAST_SHARED_ACT=1" "GVF_RAST_SHARED_ACT=0"
scripts/gpu_ab.sh $O/blend_branchy_ab.txt 3 raster "GVF_X=product" "GVF_LIB=$V/libgvf_hip_blendbr.so"
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/live_prof -o live -- python $OLDPWD/bench.py --live-only > /dev/null 2>&1 ); find /tmp/live_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/live_render_kernel_stats.csv; head -12 $O/live_render_kernel_stats.csv | cut -c1-60,200-330
