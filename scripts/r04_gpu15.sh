#!/bin/bash
# live render job: per-kernel time (rocprofv3 --kernel-trace --stats) + the plain line
set -x
mkdir -p gpurun_out/r04live
export TMPDIR=/tmp
R=$PWD
python bench.py --live-only > gpurun_out/r04live/live_line.json 2> gpurun_out/r04live/live_err.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_live -o live -- python $R/bench.py --live-only > $R/gpurun_out/r04live/live_prof_line.json 2> $R/gpurun_out/r04live/prof_err.log
cd $R
f=$(find /tmp/prof_live -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r04live/live_kernel_stats.csv
head -30 gpurun_out/r04live/live_kernel_stats.csv | cut -c1-200
cat gpurun_out/r04live/live_line.json
