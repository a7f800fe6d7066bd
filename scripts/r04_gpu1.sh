#!/bin/bash
# round 4, first GPU session: micro-benchmarks that decide the attention / row-block work, the in-flight capture reproducer, measured chain numbers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
( scripts/ubench/solo_wave.bin > $O/solo_wave.txt 2>&1 )
for v in base o2 solo solo_o2 solo_o1; do echo "== $v" >> $O/xt_variants.txt; timeout 120 scripts/ubench/xt_$v.bin 30 1 >> $O/xt_variants.txt 2>&1; done
for v in base o2 solo solo_o2; do echo "== $v (repeat)" >> $O/xt_variants.txt; timeout 120 scripts/ubench/xt_$v.bin 30 1 >> $O/xt_variants.txt 2>&1; done
for c in 1 3; do for m in 0 1 2 3 4 5; do echo "== case $c RB_COLD=$m" >> $O/rb_cold.txt; if [ $m = 0 ]; then timeout 120 scripts/ubench/rb_base.bin 50 $c >> $O/rb_cold.txt 2>&1; else RB_COLD=$m timeout 120 scripts/ubench/rb_base.bin 50 $c >> $O/rb_cold.txt 2>&1; fi; done; done
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -s -k "full_size" > $O/pipeline_full.txt 2>&1
REPRO_ROUNDS=40 timeout 900 python scripts/inflight_capture_repro.py > $O/repro_default.txt 2>&1
tail -3 $O/repro_default.txt
