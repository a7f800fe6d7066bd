# round-4 closing evidence after the tile-sort change (rasteriser sources changed: PMC summaries re-stamped BEFORE the bench line is taken)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
scripts/gpu_profile.sh r04c --steps 10 --warmup 2 --no-cpu-baseline --no-dit --streams 1 > $O/profile.log 2>&1
cp gpurun_out/prof_r04c/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
scripts/gpu_pmc.sh fetch6 "FETCH_SIZE" --steps 3 --warmup 1 --no-cpu-baseline --no-dit --streams 1 > /dev/null 2>&1
scripts/gpu_pmc.sh write6 "WRITE_SIZE" --steps 3 --warmup 1 --no-cpu-baseline --no-dit --streams 1 > /dev/null 2>&1
F=$(ls gpurun_out/pmc_fetch6/*counter_collection.csv | head -1); Wf=$(ls gpurun_out/pmc_write6/*counter_collection.csv | head -1)
cp $F $O/pmc_fetch_counter_collection.csv; cp $Wf $O/pmc_write_counter_collection.csv
python scripts/pmc_summary.py $F $Wf $O/pmc_raster.json 24 > $O/pmc_raster_summary.txt
scripts/pmc_py.sh rast6 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dit --streams 1 > /dev/null 2>&1
python scripts/pmc_sq_summary.py $O/pmc_raster_sq_summary.txt $O/pmc_raster_sq.json gpurun_out/pmc_rast6/pass1.csv gpurun_out/pmc_rast6/pass2.csv gpurun_out/pmc_rast6/pass3.csv > /dev/null 2>&1
cp $O/pmc_raster.json profiles/r04c_pmc_raster.json; cp $O/pmc_raster_sq.json profiles/r04c_pmc_raster_sq.json
python bench.py 2>$O/bench_err.log | tail -1 > $O/bench_line.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.txt
rm -rf gpurun_out/pmc_*/pass*.csv gpurun_out/prof_r04c/*kernel_trace.csv
ls -la $O; head -c 700 $O/bench_line.json; echo; tail -3 $O/smoke.txt; cat $O/pytest_gpu.txt
