"""Per-kernel summary of SQ-counter passes (rocprofv3 --pmc ... --kernel-trace, counter_collection CSVs of one or more passes).
    python scripts/pmc_sq_summary.py <out.txt> <out.json|-> <pass1.csv> [<pass2.csv> ...]
Per kernel (and grid size): dispatch count, mean duration (from the dispatch timestamps, under the profiler), the SQ_WAVE_CYCLES
fractions WAIT_ANY (parked on s_waitcnt / barrier), WAIT_INST_ANY (issue stalled), ACTIVE_INST_ANY (issuing),
  valu_issue = SQ_INSTS_VALU x 2 cycles (a wave64 instruction on a SIMD-32) / (duration x 1024 SIMDs x clock): share of the VALU
               pipe cycles in use (MFMA and transcendental instructions are counted once each; the measured ceiling of plain
               VALU code is ~0.75-0.8 of this scale: one instruction per 2.5-2.7 cycles per SIMD, profiles/r01f_ubench_valu_rates.txt),
  mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 1024 SIMDs x clock),
  clock      = GRBM_GUI_ACTIVE / 8 XCDs / duration when that pass is present and the launch is long enough for the ratio to mean
               something (>= 40 us, 1.0-2.45 GHz), else 2.1 GHz.
The JSON form is stamped with the rasteriser source hash (bench.py reads `valu_issue` of blend_kernel from it)."""
import collections, csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

out_txt, out_json, paths = sys.argv[1], sys.argv[2], sys.argv[3:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in paths:
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        m = re.search(r"(\w+)(<[^(]*>)?\(", name)
        short = (m.group(1) + (m.group(2) or "")) if m else name[:48]
        key = (short, int(r["Grid_Size"]))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
rows = []
for key, c in agg.items():
    mean = lambda n: (sum(c[n]) / len(c[n])) if n in c and c[n] else None
    d = sum(dur[key]) / len(dur[key])
    wc = mean("SQ_WAVE_CYCLES")
    gui = mean("GRBM_GUI_ACTIVE")
    clock = gui / 8.0 / (d * 1e3) if gui else 2.1                # GHz (cycles per ns); the counter sums the 8 XCDs
    if d < 40.0 or not (1.0 <= clock <= 2.45):
        clock = 2.1
    cyc = d * 1e3 * clock * 1024.0                               # SIMD-cycles of the launch
    valu_issue = (mean("SQ_INSTS_VALU") or 0) * 2.0 / cyc
    mfma = mean("SQ_VALU_MFMA_BUSY_CYCLES")
    rows.append(dict(kernel=key[0], grid=key[1], n=len(dur[key]), dur_us=round(d, 1), clock_ghz=round(clock, 3),
                     wait_any=None if not wc else round((mean("SQ_WAIT_ANY") or 0) / wc, 3),
                     wait_inst_any=None if not wc else round((mean("SQ_WAIT_INST_ANY") or 0) / wc, 3),
                     active_inst_any=None if not wc else round((mean("SQ_ACTIVE_INST_ANY") or 0) / wc, 3),
                     wait_inst_lds=None if not wc else round((mean("SQ_WAIT_INST_LDS") or 0) / wc, 3),
                     insts_valu=mean("SQ_INSTS_VALU"), insts_mfma=mean("SQ_INSTS_MFMA"), insts_lds=mean("SQ_INSTS_LDS"),
                     valu_issue=round(valu_issue, 3), mfma_busy_cycles=mfma, mfma_util=None if mfma is None else round(mfma / cyc, 3)))
rows.sort(key=lambda r: -r["dur_us"] * r["n"])
with open(out_txt, "w") as f:
    f.write("# " + (__doc__ or "").strip().replace("\n", "\n# ") + "\n")
    for r in rows:
        if r["dur_us"] * r["n"] < 20:
            continue
        f.write(f"{r['kernel'][:44]:44s} grid {r['grid']:8d} n={r['n']:4d} dur_us {r['dur_us']:8.1f} clk {r['clock_ghz']:.2f}  WAIT_ANY {r['wait_any']}  "
                f"WAIT_INST_ANY {r['wait_inst_any']}  ACTIVE {r['active_inst_any']}  WAIT_LDS {r['wait_inst_lds']}  valu_issue {r['valu_issue']}  "
                f"mfma_util {r['mfma_util']}\n")
print(open(out_txt).read())
if out_json != "-":
    from gvfdiffusion_amd._build import raster_source_hash
    doc = {"source": "rocprofv3 --pmc <SQ counters> --kernel-trace (separate passes, scripts/pmc_bin.sh / pmc_attn.sh); scripts/pmc_sq_summary.py",
           "raster_source_hash": raster_source_hash(), "kernels": {}}
    for r in rows:
        doc["kernels"].setdefault(r["kernel"], {k: r[k] for k in ("dur_us", "clock_ghz", "valu_issue", "wait_any", "wait_inst_any", "active_inst_any", "mfma_util")})
    json.dump(doc, open(out_json, "w"), indent=1)
