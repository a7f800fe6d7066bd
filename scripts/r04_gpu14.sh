#!/bin/bash
# DiT breakdown per dtype only (the divisor of scripts/dit_breakdown.py is now inferred from the trace)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export GVF_BENCH_DIT_CFG3=0 GVF_BENCH_DIT_INFLIGHT=0 GVF_BENCH_DIT_OTHER_DTYPE=0
for t in fp16 bf16; do
  GVF_DIT_DTYPE=$t scripts/gpu_profile.sh dit_$t --dit-only > /dev/null 2>&1
  python scripts/dit_breakdown.py gpurun_out/prof_dit_$t/dit_${t}_kernel_trace.csv auto > gpurun_out/r04/dit_kernel_breakdown_$t.txt
  cp gpurun_out/prof_dit_$t/dit_${t}_kernel_stats.csv gpurun_out/r04/dit_kernel_stats_$t.csv
  rm -f gpurun_out/prof_dit_$t/*kernel_trace.csv
  cat gpurun_out/r04/dit_kernel_breakdown_$t.txt
done
