"""Timeline of a two-streams-in-flight rasteriser run from a rocprofv3 kernel trace: how long each stage's kernels take when another sample's
stages run beside them, and how much of the wall time has 1 / 2 kernels in flight.   usage: python scripts/rast_overlap_trace.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    short = next((k for k in ("blend_kernel", "preprocess_kernel", "tile_sort_kernel", "bin_kernel", "morton", "bbox", "seg_", "upload_frames") if k in n), None)
    if short is None:
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Stream_Id", r.get("Queue_Id", "?"))))
ev.sort()
# keep the last 60 % (the timed two-stream loop comes last)
t_lo = ev[int(len(ev) * 0.6)][0]
ev = [e for e in ev if e[0] >= t_lo]
dur = collections.defaultdict(list)
for s, e, k, q in ev:
    # how much of this kernel's life overlapped a blend of the other stream
    ov = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, k2, q2 in ev if k2 == "blend_kernel" and (s2, e2) != (s, e))
    dur[k].append(((e - s) / 1e3, ov / max(1, e - s)))
for k, v in sorted(dur.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
    alone = [d for d, o in v if o < 0.1]
    shared = [d for d, o in v if o > 0.6]
    print(f"{k:20s} n={len(v):4d} avg {sum(d for d, _ in v) / len(v):8.1f} us | alone (<10 % beside a blend) n={len(alone):3d} avg {sum(alone) / max(1, len(alone)):8.1f}"
          f" | beside a blend (>60 %) n={len(shared):3d} avg {sum(shared) / max(1, len(shared)):8.1f}")
# concurrency histogram
pts = sorted([(s, 1) for s, e, k, q in ev] + [(e, -1) for s, e, k, q in ev])
lvl, last, hist = 0, pts[0][0], collections.Counter()
for t, d in pts:
    hist[lvl] += t - last
    last, lvl = t, lvl + d
tot = sum(hist.values())
print("kernels in flight: " + "  ".join(f"{k}: {v / tot * 100:.1f} %" for k, v in sorted(hist.items())))
