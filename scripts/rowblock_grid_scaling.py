"""Is the row-block launch bound by each CU's own fill path or by the XCD's L2 -> CU fabric?  Same kernel, same rows per workgroup, fewer
workgroups (blockIdx round-robins over the 8 XCDs, so n workgroups = n / 8 active CUs per XCD): a per-CU bound gives the same time per
launch for any n <= 256, a shared-fabric bound gives time ~ n."""
import math
import sys
import torch

from gvfdiffusion_amd.ops import dit_ops

dev = torch.device("cuda:0")
C = 512
lp = torch.float16
g = torch.Generator().manual_seed(0)
w1 = (torch.randn((C, C), generator=g) / math.sqrt(C)).to(lp).to(dev)
f1 = (torch.randn((2048, C), generator=g) / math.sqrt(C)).to(lp).to(dev)
f2 = (torch.randn((C, 2048), generator=g) / math.sqrt(2048)).to(lp).to(dev)
w3 = (torch.randn((3 * C, C), generator=g) / math.sqrt(C)).to(lp).to(dev)
lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
for mlp in (False, True):
    stream = dit_ops.rowblock_pack_stream(w1, mlp=(f1, f2) if mlp else None, w3=w3)
    for nblk in (32, 64, 128, 192, 256, 384, 512):
        M = nblk * 48
        a = torch.randn((M, C), device=dev).to(lp)
        x = torch.randn((M, C), device=dev)
        out = torch.empty((M, 3 * C), dtype=lp, device=dev)
        kw = dict(b1=lb, ln1=dict(ln_w=lw, ln_b=lb), out3=out, b3=torch.zeros(3 * C, device=dev))
        if mlp:
            kw.update(mlp_bias=(torch.zeros(2048, device=dev), lb), hidden=2048, ln2=dict(ln_w=lw, ln_b=lb))
        for _ in range(5):
            dit_ops.rowblock_fused(a, stream, x, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 50
        for _ in range(n):
            dit_ops.rowblock_fused(a, stream, x, **kw)
        e1.record(); torch.cuda.synchronize()
        print(f"mlp={int(mlp)} workgroups {nblk:4d}: {e0.elapsed_time(e1) / n * 1e3:7.1f} us per launch")
