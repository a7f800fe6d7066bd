"""How many per-tile segments does the distribution sort hand to the sorting network (a bucket of more than BKT_MAX_RUN keys)?  Needs the SORT_STATS variant:
    python -m gvfdiffusion_amd._build --variant sortstats rast.hip=-DSORT_STATS
    GVF_LIB=gvfdiffusion_amd/variants/libgvf_hip_sortstats.so python scripts/sort_stats.py [live|bench]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gvfdiffusion_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
fn = L.gvf_debug_sort_stats
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
out = (ctypes.c_uint64 * 16)()
which = sys.argv[1] if len(sys.argv) > 1 else "live"
if which == "live":
    os.environ["GVF_LIVE_STREAMS"] = "1"
    fn(None, 1)
    r = bench.bench_live_render(dev)
    print("live job:", r["ms_per_sample"], "ms per sample")
else:
    w = bench.RasterWorkload(dev, 262144, 800, 24, 2, 0)
    fn(None, 1)
    w.step(); torch.cuda.synchronize()
fn(out, 0)
v = list(out)
print(f"{which}: distribution-sort segments {v[0]}, sent to the network {v[1]} ({100.0 * v[1] / max(v[0], 1):.2f} %); keys {v[2]}, of them in network segments {v[3]} ({100.0 * v[3] / max(v[2], 1):.2f} %)")
print("longest bucket histogram (0-7, 8-15, ..., 88+):", v[4:16])
