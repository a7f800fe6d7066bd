"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one or more passes): kernel, grid -> counter means."""
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        short = name.split("(")[0].split("::")[-1][:40]
        if "<" in name and "kernel<" in name:
            short += "<" + name.split("kernel<")[1].split(">")[0] + ">"
        key = (short, r.get("Grid_Size", r.get("Grid_Size_X", "?")))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, ctrs in sorted(agg.items()):
    n = max(len(v) for v in ctrs.values())
    if n < 3:
        continue
    print(f"== {key[0]} grid={key[1]} dispatches={n}")
    wc = sum(ctrs.get("SQ_WAVE_CYCLES", [0])) / max(1, len(ctrs.get("SQ_WAVE_CYCLES", [0])))
    for c, v in sorted(ctrs.items()):
        m = sum(v) / len(v)
        extra = f"  ({m / wc:.3f} of WAVE_CYCLES)" if wc and c.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES")) else ""
        print(f"   {c:28s} {m:16.0f}{extra}")
