#!/bin/bash
# attn_xt64 in the product: VAE tests, decode timing both ways, both operand types
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04m; mkdir -p $O
python -c "import gvfdiffusion_amd._build as b; b.build(verbose=False)" >/dev/null 2>&1
timeout 1500 python -m pytest tests/test_vae_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_vae.txt
for r in 1; do
  echo "tiled64:" >> $O/vae_ab.txt; python scripts/vae_breakdown.py >> $O/vae_ab.txt 2>&1
  echo "tiled64, GEMM:" >> $O/vae_ab.txt; GVF_VAE_FOLD=0 python scripts/vae_breakdown.py >> $O/vae_ab.txt 2>&1
  echo "kvres:" >> $O/vae_ab.txt; GVF_VAE_TILED64=0 python scripts/vae_breakdown.py >> $O/vae_ab.txt 2>&1
done
grep -v amdgpu.ids $O/vae_ab.txt
