"""Per-kernel / per-grid breakdown of a rocprofv3 kernel trace of `bench.py --dit-only` (36 NFEs)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nfe = float(sys.argv[2]) if len(sys.argv) > 2 else 36.0
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    short = 'attn' if 'attn_fwd' in n else ('gemm' + n.split('gemm_bf16_kernel<')[1].split('>')[0] if 'gemm_bf16_kernel' in n else ('ln' if 'ln_mod' in n else None))
    if 'attn_fwd' in n:
        short += '<T>' if 'true' in n else ''
    if short:
        agg[(short, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / nfe / 1e6 > 0.04:
        print(k, len(v), 'avg %.1f us' % (sum(v) / len(v) / 1e3), 'min %.1f' % (min(v) / 1e3), 'max %.1f' % (max(v) / 1e3), 'per NFE %.2f ms' % (sum(v) / nfe / 1e6))
    tot += sum(v)
print('sum per NFE %.2f ms' % (tot / nfe / 1e6))
