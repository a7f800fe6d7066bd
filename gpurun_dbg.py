import sys, os, subprocess
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch
from gvfdiffusion_amd.ops import dit_ops
from oracle import dit_ref
cuda = torch.device('cuda:0')
def bf(x): return x.to(torch.bfloat16)
for (N, Lq, Lk, H) in [(3, 200, 77, 2), (1, 32, 32, 1), (1, 32, 20, 1), (1, 32, 64, 1), (1,32,70,1), (1, 130, 1370, 4)]:
    g = torch.Generator().manual_seed(N * 1000 + Lq + Lk)
    q = bf(torch.randn((N, Lq, H, 32), generator=g) * 2).to(cuda)
    k = bf(torch.randn((N, Lk, H, 32), generator=g) * 2).to(cuda)
    v = bf(torch.randn((N, Lk, H, 32), generator=g)).to(cuda)
    out = torch.empty_like(q)
    sq, sk = (Lq * H * 32, 0, H * 32), (Lk * H * 32, 0, H * 32)
    dit_ops.attention_bf16(q, k, v, out, N, 1, Lq, Lk, H, sq, sk, sk, sq, None, None)
    ref = dit_ref.sdpa(q.float(), k.float(), v.float(), "bf16")
    nan = torch.isnan(out.float())
    d = (out.float() - ref).abs()
    print((N, Lq, Lk, H), "nan count", int(nan.sum()), "of", out.numel(), "max diff (non-nan)", float(d[~nan].max()) if (~nan).any() else None,
          "nan rows (q idx)", sorted(set(torch.nonzero(nan)[:, 1].tolist()))[:10], "nan d", sorted(set(torch.nonzero(nan)[:, 3].tolist()))[:40])
