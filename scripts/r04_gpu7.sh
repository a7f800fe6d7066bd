#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04g; mkdir -p $O
run() { echo "== $*" >> $O/which_stage_bisect.txt; env "$@" REPRO_ROUNDS=5 timeout 200 python scripts/inflight_which_stage.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-300 >> $O/which_stage_bisect.txt; }
run REPRO_KINDS=vae_latents,kvres64,gemm
run REPRO_KINDS=vae_decode GVF_DIT_PREFETCH=0
run REPRO_KINDS=vae_decode GVF_DIT_ROWBLOCK=0
run REPRO_KINDS=vae_decode GVF_DIT_TEMPORAL_FUSED=0
run REPRO_KINDS=vae_decode GVF_DIT_TILED_KV=0
run REPRO_KINDS=vae_decode GVF_DIT_DTYPE=bf16
cat $O/which_stage_bisect.txt
