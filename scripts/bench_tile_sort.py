"""Times gvf_tile_sort_u64 (the per-tile half of the rasteriser's sort) on synthetic segments of one size each, depths uniform over a
range as in the bench scene.  python scripts/bench_tile_sort.py [lib.so]  -- a second library (e.g. built with -DSORT_BUCKETS=0, the
round-1 sorting networks) can be passed to compare."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gvfdiffusion_amd import _build

path = sys.argv[1] if len(sys.argv) > 1 else _build.LIB_PATH
lib = ctypes.CDLL(path)
lib.gvf_tile_sort_u64.restype = ctypes.c_int
lib.gvf_tile_sort_u64.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
print(os.path.basename(path))
for n, nseg in [(300, 20000), (1100, 16000), (1800, 8000), (3000, 4000), (6000, 2000), (12000, 1000), (16000, 800)]:
    d = rng.uniform(0.8, 1.6, n * nseg).astype(np.float32)
    ids = np.tile(rng.permutation(262144)[:n].astype(np.uint64), nseg)
    k = (d.view(np.uint32).astype(np.uint64) << np.uint64(32)) | ids
    keys = torch.from_numpy(k.view(np.int64)).to(dev)
    ranges = torch.tensor([[i * n, (i + 1) * n] for i in range(nseg)], dtype=torch.int32, device=dev)
    out = torch.empty(n * nseg, dtype=torch.int32, device=dev)
    scratch = torch.empty(2 + 2 * nseg, dtype=torch.int32, device=dev)
    work = keys.clone()
    def run():
        work.copy_(keys)                          # the global class sorts in place; keep the input the same every time
        rc = lib.gvf_tile_sort_u64(work.data_ptr(), ranges.data_ptr(), nseg, out.data_ptr(), scratch.data_ptr(), None)
        assert rc == 0
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5): work.copy_(keys)
    torch.cuda.synchronize()
    t_copy = (time.perf_counter() - t0) / 5
    want = (np.sort(k[:n]) & np.uint64(0xffffffff)).astype(np.uint32)
    assert np.array_equal(out[:n].cpu().numpy().view(np.uint32), want)
    dt = t_all - t_copy
    print(f"  {nseg:6d} segments of {n:6d} keys: {dt * 1e3:8.3f} ms  = {n * nseg / dt / 1e9:6.2f} G keys/s")
