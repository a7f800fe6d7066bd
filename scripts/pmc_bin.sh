#!/bin/bash
# usage: scripts/pmc_bin.sh <tag> <command...> : SQ counter passes of a standalone benchmark, summarised per kernel
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
CMD0=$(realpath "$1"); shift
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES" \
            "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${TAG}_$i
  timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$i -o p$i -- "$CMD0" "$@" > "$OUT/log$i.txt" 2>&1 < /dev/null
  echo "pass $i rc=$?"
  for f in $(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv" 2>/dev/null); do cp "$f" "$OUT/pass$i.csv"; done
done
python $REPO/scripts/pmc_kernel_summary.py "$OUT"/pass*.csv | tee "$OUT/summary.txt"
