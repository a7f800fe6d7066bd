#!/bin/bash
# builds scripts/ubench/xt_<name>.bin for every "name:flags" argument (attn_xt_bench.hip with extra -D flags)
HERE=$(cd "$(dirname "$0")" && pwd)
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -fno-slp-vectorize"
for v in "$@"; do
  n=${v%%:*}; fl=${v#*:}
  ( hipcc $F $fl "$HERE/attn_xt_bench.hip" -o "$HERE/xt_$n.bin" 2>&1 | grep -E " error|error:" ) &
done
wait
ls "$HERE"/xt_*.bin | wc -l
