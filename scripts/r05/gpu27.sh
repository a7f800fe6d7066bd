#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05z; mkdir -p $O
scripts/gpu_ab.sh $O/presorted_live_ab.txt 3 live "GVF_BENCH_PRESORT=0" "GVF_BENCH_PRESORT=1"
