#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05v; mkdir -p $O
timeout 1500 python -m pytest tests/test_rast_gpu.py tests/test_render_driver_gpu.py tests/test_pipeline_gpu.py tests/test_inference_script_gpu.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest_rast.txt; cat $O/pytest_rast.txt
scripts/gpu_ab.sh $O/live_slot_ab.txt 3 live "GVF_RAST_SLOT_ORDER=1" "GVF_RAST_SLOT_ORDER=0"
for s in 1 0; do echo -n "streams=1 slot=$s " >> $O/live_slot_ab.txt; GVF_RAST_SLOT_ORDER=$s GVF_LIVE_STREAMS=1 python bench.py --live-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_sample'], d['value'])" >> $O/live_slot_ab.txt; done
tail -2 $O/live_slot_ab.txt
