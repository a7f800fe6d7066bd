"""Where does the adaptive chain's time outside the DiT forwards go?  Reads a rocprofv3 --kernel-trace csv of `bench.py --e2e-only` and splits the LAST
sampling stage (from the first kernel after the last `key_order` / cache-builder launch to the first VAE kernel) into: DiT-forward kernels, solver
kernels (everything else), and idle gaps on the device, per model evaluation.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -o gt -- python bench.py --e2e-only; python scripts/adaptive_gap_trace.py /tmp/gt/*/gt_kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# the last chain: from the last attn_pack_kv launch (end of prepare_conditions) to the first attn_xt64 after it (the VAE's cross attention)
last_pack = max(i for i, e in enumerate(ev) if "attn_pack_kv_kernel" in e[2])
end = next(i for i in range(last_pack, len(ev)) if any(k in ev[i][2] for k in ("gemm8_kernel", "query_embed", "attn_xt64", "ln_mod_kernel<3")))
seg = ev[last_pack + 1:end]
fwd = lambda n: any(k in n for k in ("attn_xt_kernel", "rowblock_kernel", "timestep_embed", "modulation", "final_layer"))
t_f = sum(e[1] - e[0] for e in seg if fwd(e[2])); t_s = sum(e[1] - e[0] for e in seg if not fwd(e[2]))
n_evals = sum(1 for e in seg if "final_layer" in e[2])
gaps = [b[0] - a[1] for a, b in zip(seg, seg[1:]) if b[0] > a[1]]
wall = seg[-1][1] - seg[0][0]
print(f"sampling stage (device view): {wall / 1e6:.2f} ms, {n_evals} evaluations, {len(seg)} launches")
print(f"  DiT-forward kernels {t_f / 1e6:8.2f} ms  ({t_f / 1e3 / max(n_evals, 1):.1f} us per evaluation)")
print(f"  solver kernels      {t_s / 1e6:8.2f} ms  ({t_s / 1e3 / max(n_evals, 1):.1f} us per evaluation, {sum(1 for e in seg if not fwd(e[2])) / max(n_evals, 1):.1f} launches per evaluation)")
print(f"  idle gaps           {sum(gaps) / 1e6:8.2f} ms  ({sum(gaps) / 1e3 / max(n_evals, 1):.1f} us per evaluation; gaps > 20 us: {sum(1 for g in gaps if g > 20000)} totalling {sum(g for g in gaps if g > 20000) / 1e6:.2f} ms)")
c = collections.Counter()
for e in seg:
    if not fwd(e[2]): c[e[2][:90]] += e[1] - e[0]
for k, v in c.most_common(14): print(f"    {v / 1e3:8.1f} us  {k}")
