#!/bin/bash
# motion-VAE decode: where the 24 ms go (kernel stats of scripts/vae_breakdown.py, bf16 default)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r04o; mkdir -p $O
python -c "import gvfdiffusion_amd._build as b; b.build(verbose=False)" >/dev/null 2>&1
python scripts/vae_breakdown.py > $O/vae_breakdown.txt 2>&1
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o vae -- python $GRAFT_REPO_ROOT/scripts/vae_breakdown.py > $O/prof.log 2>&1 )
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/vae_kernel_stats.csv
rm -rf $O/prof
cat $O/vae_breakdown.txt; head -25 $O/vae_kernel_stats.csv | cut -c1-200
