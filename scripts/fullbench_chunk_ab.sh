cd /root/repo
export GVF_BENCH_SHARD_COMPARE=0
for rep in 1 2; do for V in 2147483648 134217728; do
echo -n "[chunk=$V] "; GVF_DIT_KV_CHUNK_BYTES=$V timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['end_to_end']; print(d['dit']['ms_per_nfe'], e['wall_ms'], e['stage_ms'], d['sharded_sampling']['wall_ms'])"
done; done
