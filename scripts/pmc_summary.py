"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; scripts/gpu_pmc.sh) per kernel.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section) -> doubled here.  Writes the JSON bench.py reads for
`roofline.traffic`.   usage: python scripts/pmc_summary.py <fetch.csv> <write.csv> <out.json> <frames_per_launch>"""
import collections, csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvfdiffusion_amd._build import raster_source_hash


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        m = re.search(r"(\w+)(<[^(]*>)?\(", name)
        agg[(m.group(1) + (m.group(2) or "")) if m else name[:40]].append(float(r["Counter_Value"]) * 1024.0)
    return agg


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace; bench.py --steps 3 --warmup 1 "
                 "--no-cpu-baseline --no-dit", "fetch_correction": "x2 (gfx950: 128-B requests tallied at 64 B)",
       "raster_source_hash": raster_source_hash(), "kernels": {}}
for k in fetch:
    if not any(s in k for s in ("kernel", "sort_")) or "elementwise" in k:
        continue
    # skip the first (warm-up of the allocator / first touch) launch when there are several
    f = fetch[k][1:] or fetch[k]
    w = write.get(k, [0.0])[1:] or write.get(k, [0.0])
    fb, wb = 2.0 * sum(f) / len(f), sum(w) / len(w)
    out["kernels"][k] = {"launches": len(fetch[k]), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                         "traffic_bytes_per_launch": round(fb + wb)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:28s} fetch {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {v['write_bytes_per_launch'] / 1e6:9.1f} MB")
