"""Where the DiT step's time outside its kernels goes: the graphed forward alone, back to back, against the sampler's per-NFE time (GPU only)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
w = bench.DiTWorkload(dev)
w.sample(steps=4); w.sample(steps=32)
torch.cuda.synchronize(); t0 = time.perf_counter(); w.sample(steps=32); torch.cuda.synchronize()
print("sampler, 32 NFE: %.3f ms per NFE" % ((time.perf_counter() - t0) / 32 * 1e3))
t = torch.tensor([500.0], device=dev)
kw = dict(w.cond)
for _ in range(3): y = w.model(w.x, t, **kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(32): y = w.model(w.x, t, **kw)
torch.cuda.synchronize()
print("forward only (graph replay + input copies), 32 calls: %.3f ms per call" % ((time.perf_counter() - t0) / 32 * 1e3))
g = getattr(w.model, "_graph", None)
if g is not None and "graph" in g:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(32): g["graph"].replay()
    torch.cuda.synchronize()
    print("graph.replay() only, 32 calls: %.3f ms per call" % ((time.perf_counter() - t0) / 32 * 1e3))
# host side of the sampler: time until sample() returns (everything enqueued) against the time until the GPU is done
torch.cuda.synchronize(); t0 = time.perf_counter(); w.sample(steps=32); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("sampler: host returns after %.2f ms, GPU done after %.2f ms (32 NFE)" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); w.sample(steps=32); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
