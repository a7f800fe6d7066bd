"""DiT denoise step at frame counts other than the bench's T = 24 (B = 1, configs/diffusion.yml): the row-block path pads a sample's rows
to whole 48-row blocks when T * 512 is not a multiple of 48 (T = 16, 32); GVF_DIT_ROWBLOCK=0 gives the per-sub-layer launches.
python scripts/dit_shapes.py [T ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda:0")
for T in [int(a) for a in sys.argv[1:]] or [16, 24, 32]:
    line = f"T={T:3d} ({T * 512} rows)"
    for rb in (True, False):
        w = bench.DiTWorkload(dev, T=T)
        w.model.use_rowblock = rb
        w.sample(steps=4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w.sample(steps=16)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 16 * 1e3
        line += f"   {'row-block' if rb else 'per-sub-layer'}: {ms:6.3f} ms / NFE ({w.flops_per_nfe(True) / ms / 1e9:6.1f} TFLOP/s)"
        del w
        torch.cuda.empty_cache()
    print(line)
