#!/bin/bash
# One A/B/A/... measurement on ONE box (boxes of the pool differ by up to 10 %: never compare across gpurun calls), parametrised
# (replaces the 18 one-shot scripts/r04_gpu*.sh of round 4):
#   scripts/gpu_ab.sh <out-file> <reps> <leg> "<ENV settings of variant A>" "<ENV settings of variant B>" [...]
#   leg: dit (ms_per_nfe) | raster (ms_per_step_serial + blend / preprocess / tile_sort stage ms) | live (ms_per_sample) | e2e (wall_ms + stages)
# e.g.  scripts/gpu_ab.sh gpurun_out/r05/prefetch_ab.txt 3 dit "GVF_DIT_PREFETCH=1" "GVF_DIT_PREFETCH=0"
# A variant may rebuild the library first:  "REBUILD='-DXT_PERSIST=0' ..."  is NOT supported here on purpose -- build the variants as
# separate .so files before the call and select them with GVF_LIB=<path> (gvfdiffusion_amd/_lib.py).
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=${1:?out}; REPS=${2:?reps}; LEG=${3:?leg}; shift 3
mkdir -p "$(dirname "$OUT")"
export GVF_BENCH_DIT_CFG3=0 GVF_BENCH_DIT_INFLIGHT=0 GVF_BENCH_DIT_OTHER_DTYPE=0 GVF_BENCH_DIT_HOSTILE=${GVF_BENCH_DIT_HOSTILE:-0}
case $LEG in
  dit)    ARGS="--dit-only --no-cpu-baseline"; PY="print(d['ms_per_nfe'], d['value'], d.get('softmax_guard'))" ;;
  raster) ARGS="--no-dit --no-cpu-baseline --streams 1"; PY="s=d['stage_ms_per_step']; print(d['ms_per_step_serial'], s['blend'], s['preprocess'], s['tile_sort'], d['value'])" ;;
  live)   ARGS="--live-only"; PY="print(d['ms_per_sample'], d['value'])" ;;
  e2e)    ARGS="--e2e-only"; PY="print(d['wall_ms'], d['stage_ms'], d['nfe'])" ;;
  *) echo "unknown leg $LEG"; exit 2 ;;
esac
for rep in $(seq 1 $REPS); do
  for V in "$@"; do
    echo -n "[$V] " >> "$OUT"
    env $V timeout 900 python bench.py $ARGS 2>>"$OUT.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); $PY" >> "$OUT" 2>&1
  done
done
cat "$OUT"
