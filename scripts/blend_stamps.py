"""Wave-cycles per phase of blend_kernel on the bench shape (BASELINE configs[1]) -- needs the BLEND_TIMING variant library:
    python -m gvfdiffusion_amd._build --variant blendt rast.hip=-DBLEND_TIMING
    GVF_LIB=gvfdiffusion_amd/variants/libgvf_hip_blendt.so python scripts/blend_stamps.py
Forcing s_waitcnt vmcnt(0) between the phases perturbs the schedule (the product overlaps the id load, the gather and the staging
arithmetic of one wave only through other waves anyway); the figures are shares of the waves' resident time, not product timings."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gvfdiffusion_amd import _lib

dev = torch.device("cuda:0")
P, S, F = int(os.environ.get("P", 262144)), int(os.environ.get("S", 800)), int(os.environ.get("F", 24))
w = bench.RasterWorkload(dev, P, S, F, 2, 0)
L = _lib.lib()
fn = L.gvf_debug_blend_timing
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
tiles = ((S + 15) // 16) ** 2
rows = F * tiles * 4
buf = torch.zeros((rows, 12), dtype=torch.int64, device=dev)
for _ in range(2): w.step()
torch.cuda.synchronize()
assert fn(ctypes.c_void_p(buf.data_ptr()), rows) == 0
N = 1
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): w.step()
e1.record(); torch.cuda.synchronize()
b = buf.double()
v = b.sum(0).tolist()
names = ["wait@barrier1", "id load", "record gather", "stage arithmetic", "wait@barrier2", "compaction", "compositing", "rounds", "list entries", "waves", "lifetime", "epilogue stores"]
life = v[10]
print(f"step {e0.elapsed_time(e1) / N:.3f} ms (timing build), {w.D_binned} binned instances, {F} frames, waves {v[9]:.0f}")
for i in (0, 1, 2, 3, 4, 5, 6, 11):
    print(f"  {names[i]:18s} {v[i] / life * 100:6.2f} % of wave lifetime   {v[i] / v[9]:9.0f} cycles per wave")
print(f"  rounds per wave {v[7] / v[9]:.2f}, list entries per wave {v[8] / v[9]:.1f}, cycles per list entry in compositing {v[6] / max(v[8], 1):.1f}, lifetime per wave {life / v[9]:.0f} cycles")
print(f"  sum of wave lifetimes / (1024 SIMDs) = {life / 1024:.0f} cycles per SIMD-slot-sum -> at 8 waves per SIMD {life / 1024 / 8:.0f} cycles of launch")

# distribution over waves: who is slow?
life_w = b[:, 10]
q = torch.quantile(life_w, torch.tensor([0.1, 0.5, 0.9, 0.99, 1.0], dtype=torch.float64, device=dev)).tolist()
print("  wave lifetime quantiles (cycles) 10/50/90/99/100 %:", [int(x) for x in q])
for i in (1, 2, 4, 6):
    q = torch.quantile(b[:, i], torch.tensor([0.5, 0.9, 0.99], dtype=torch.float64, device=dev)).tolist()
    print(f"  {names[i]:18s} per wave 50/90/99 %: {[int(x) for x in q]}")

# How much of the launch is tail?  List-scheduling simulation on the measured workgroup durations (a workgroup = the longest of its 4 waves; 2048 resident
# slots = 8 per CU; durations are what the waves took while sharing their SIMD with 7 others, so this prices the ORDER only): makespan in dispatch order
# (blockIdx.x = tile fastest, then frames), heaviest first, and the bound sum / slots.
import heapq
import numpy as np
wg = b[:, 10].reshape(-1, 4).max(1).values.cpu().numpy()          # [frames * tiles] in dispatch order (f major, tile minor)
def makespan(d, slots=2048):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for x in d:
        t = heapq.heappop(h) + float(x)
        end = max(end, t)
        heapq.heappush(h, t)
    return end
ideal = wg.sum() / 2048
print(f"  schedule simulation (cycles): ideal {ideal:.0f}, dispatch order {makespan(wg):.0f} (+{(makespan(wg) / ideal - 1) * 100:.1f} %), "
      f"heaviest first {makespan(np.sort(wg)[::-1]):.0f} (+{(makespan(np.sort(wg)[::-1]) / ideal - 1) * 100:.1f} %), longest workgroup {wg.max():.0f}")
cnt = b[:, 8].reshape(-1, 4).sum(1).cpu().numpy()
order = np.argsort(-cnt, kind="stable")                        # by list entries (known before the launch would be: instances per tile)
print(f"  ... ordered by list entries per tile (a proxy the device has before the launch): {makespan(wg[order]):.0f} (+{(makespan(wg[order]) / ideal - 1) * 100:.1f} %)")
