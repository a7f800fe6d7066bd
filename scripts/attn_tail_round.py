"""Does the half-empty last round of the tiled attention's grid cost anything?  The static cross attention of the DiT (4096 keys shared by the
frames, 16 heads, 512 queries per frame) launches 2 workgroups of 256 queries per (frame, head): T frames -> 32 T workgroups on the 512 resident
slots of the chip (2 per CU).  T = 16 / 32 / 48 fill whole rounds, T = 24 / 40 leave a half-empty last round.  If time per frame is the same for
all of them, the tail costs nothing: the kernel is bound by the VALU / transcendental pipe, and a workgroup that has its SIMDs to itself runs
that much faster.  Prints us per launch, us per frame, for bf16 and fp16 tiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvfdiffusion_amd.ops import dit_ops   # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, H, C, Lk = 512, 16, 512, 4096


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dt in (torch.bfloat16, torch.float16):
    kv = torch.randn((Lk, 2 * C), generator=g).to(dev)
    kt, vt = dit_ops.attention_pack_kv(kv, 1, Lk, H, 0, C, dtype=dt)
    for T in (8, 16, 24, 32, 40, 48, 64):
        q = torch.randn((T * N, C), generator=g).to(dt).to(dev)
        out = torch.empty_like(q)
        st = (T * N * C, N * C, C)
        us = timeit(lambda: dit_ops.attention_tiled(q, kt, vt, out, 1, T, N, Lk, H, st, st, 1, 0))
        wgs = 2 * T * H
        print(f"{str(dt):16s} T={T:3d}  workgroups {wgs:5d} = {wgs / 512:.2f} rounds  {us:8.1f} us / launch  {us / T:6.2f} us / frame  "
              f"{4.0 * T * N * Lk * C / us / 1e6:7.1f} TFLOP/s")
