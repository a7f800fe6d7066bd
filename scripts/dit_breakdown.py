"""Per-kernel / per-grid breakdown of a rocprofv3 kernel trace of `bench.py --dit-only`.  The number of forwards in the trace is the second
argument, or (default / "auto") the number of tiled-attention launches / 36 (3 per block x 12 blocks of configs/diffusion.yml)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n_attn = sum('attn_xt_kernel' in r['Kernel_Name'] for r in rows)
nfe = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "auto" else (n_attn / 36.0 if n_attn else 36.0)


def short_name(n):
    if 'gemm_lp_kernel' in n:
        return 'gemm<' + n.split('gemm_lp_kernel<')[1].split('>')[0] + '>'
    for key in ('rowblock_kernel<true', 'rowblock_kernel<false', 'rowblock_pack', 'attn_xt_kernel<0', 'attn_xt_kernel<1', 'attn_xt_kernel', 'attn_pack_kv', 'attn_fwd_kernel', 'attn_small', 'attn_kvres', 'ln_mod', 'cast_pad', 'gemm_ln'):
        if key in n:
            return key + ('<T>' if key == 'attn_fwd_kernel' and 'true' in n else '')
    return n.split('(')[0][-40:]


agg = collections.defaultdict(list)
for r in rows:
    agg[(short_name(r['Kernel_Name']), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot, launches = 0, 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / nfe / 1e6 > 0.03:
        print(f"{k[0]:28s} grid {k[1]:6d}  n/NFE {len(v) / nfe:6.1f}  avg {sum(v) / len(v) / 1e3:7.1f} us  min {min(v) / 1e3:7.1f}  per NFE {sum(v) / nfe / 1e6:5.2f} ms")
    tot += sum(v); launches += len(v)
print('sum per NFE %.2f ms, %.0f launches per NFE (%g forwards in the trace)' % (tot / nfe / 1e6, launches / nfe, nfe))
