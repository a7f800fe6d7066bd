#!/bin/bash
# builds scripts/ubench/x64_<name>.bin for every "name:flags" argument (x64_bench.hip with extra flags)
HERE=$(cd "$(dirname "$0")" && pwd)
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -fno-honor-nans -fno-slp-vectorize"
for v in "$@"; do
  n=${v%%:*}; fl=${v#*:}
  ( hipcc $F $fl -Rpass-analysis=kernel-resource-usage "$HERE/x64_bench.hip" -o "$HERE/x64_$n.bin" 2>&1 | grep -E "error|attn_xt64_kernel" -A7 | grep -E "error|Name|VGPRs:|AGPRs|Scratch" | sed "s/^.*remark: /$n: /" ) &
done
wait
ls "$HERE"/x64_*.bin | wc -l
