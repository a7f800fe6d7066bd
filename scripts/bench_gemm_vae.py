"""Times gvf_gemm on the motion VAE's latent-block shapes (M = 24 x 512 rows, dim 768, GEGLU FF) next to torch.matmul (hipBLASLt); GPU only.
GVF_GEMM_BM=64/128 and GVF_GEMM_BK=32/64 force the tile shape (tuning aids of csrc/gemm.hip); GVF_GEMM8=0 keeps the eligible shapes off the
eight-wave 256-wide kernel (csrc/gemm8.hip), =2 sends every eligible shape there."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvfdiffusion_amd.ops import dit_ops
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
def rn(*s): return torch.randn(s, generator=g).to(torch.bfloat16).to(dev)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 12288
for name, N, K, epi in (("to_qkv", 2304, 768, 0), ("to_out+resid", 768, 768, 3), ("fc1", 6144, 768, 0), ("fc2+resid", 768, 3072, 3), ("dec to_q (P rows)", 768, 768, 0)):
    Mx = 262144 if name.startswith("dec") else M
    a, w = rn(Mx, K), rn(N, K)
    bias = torch.randn(N, generator=g).to(dev)
    if epi == 3:
        out = torch.randn((Mx, N), generator=g).to(dev)
        fn = lambda: dit_ops.gemm_bf16(a, w, bias, out, 3)
    else:
        out = torch.empty((Mx, N), dtype=torch.bfloat16, device=dev)
        fn = lambda: dit_ops.gemm_bf16(a, w, bias, out, epi)
    us = timeit(fn)
    b16 = bias.to(torch.bfloat16)
    us_t = timeit(lambda: torch.addmm(b16, a, w.t()))
    fl = 2.0 * Mx * N * K
    print(f"{name:18s} M={Mx:6d} N={N:5d} K={K:5d}: gvf {us:7.1f} us {fl/us/1e6:7.1f} TF/s | torch.addmm (bf16 out) {us_t:7.1f} us {fl/us_t/1e6:7.1f} TF/s")
