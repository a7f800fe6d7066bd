#!/bin/bash
# usage: scripts/pmc_py.sh <tag> <python script + args...> : four SQ / GRBM counter passes (kernel-trace only) of a python command;
# raw CSVs land in gpurun_out/pmc_<tag>/pass<i>.csv -- summarise with scripts/pmc_sq_summary.py
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${TAG}_$i
  timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$i -o p$i -- python "$REPO/$1" "${@:2}" > "$OUT/log$i.txt" 2>&1 < /dev/null
  echo "pass $i rc=$?"
  for f in $(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv" 2>/dev/null); do cp "$f" "$OUT/pass$i.csv"; done
done
