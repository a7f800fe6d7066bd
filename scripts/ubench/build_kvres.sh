#!/bin/bash
# builds scripts/ubench/kvres_<name>.bin for every "name:flags" argument (kvres_bench.hip with extra -D flags; csrc/attn.hip's own build flags)
HERE=$(cd "$(dirname "$0")" && pwd)
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -mllvm -amdgpu-sched-strategy=iterative-ilp -fno-slp-vectorize"
for v in "$@"; do
  n=${v%%:*}; fl=${v#*:}
  ( hipcc $F $fl -Rpass-analysis=kernel-resource-usage "$HERE/kvres_bench.hip" -o "$HERE/kvres_$n.bin" 2>&1 | grep -A9 "attn_kvres_kernelILi64ELb1ELi[01]" | grep -E "Name|VGPRs:|AGPRs|Scratch|Occupancy" | sed "s/^.*remark: /$n: /" ) &
done
wait
ls "$HERE"/kvres_*.bin | wc -l
