"""Times gvf_gemm_bf16 on the DiT's projection shapes next to torch.matmul (hipBLASLt) as a yardstick; GPU only."""
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
for name, N, K, epi in (("to_qkv", 1536, 512, 0), ("to_q", 512, 512, 0), ("to_out+resid", 512, 512, 3), ("fc1+gelu", 2048, 512, 1), ("fc2+resid", 512, 2048, 3)):
    a, w = rn(M, K), rn(N, K)
    bias = torch.randn(N, generator=g).to(dev)
    if epi == 3:
        out = torch.randn((M, N), generator=g).to(dev)
        gate = torch.randn((1, N), generator=g).to(dev)
        fn = lambda: dit_ops.gemm_bf16(a, w, bias, out, 3, gate=gate, gate_ld=N, rows_per_group=M)
        byts = M * K * 2 + N * K * 2 + 2 * M * N * 4
    else:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        fn = lambda: dit_ops.gemm_bf16(a, w, bias, out, epi)
        byts = M * K * 2 + N * K * 2 + M * N * 2
    us = timeit(fn)
    wt = w.t().contiguous()
    us_t = timeit(lambda: torch.matmul(a, w.t()))
    fl = 2.0 * M * N * K
    print(f"{name:14s} N={N:5d} K={K:5d}: gvf {us:7.1f} us {fl/us/1e6:7.1f} TF/s {byts/us/1e3:6.0f} GB/s | torch.matmul (bf16 out, no epilogue) {us_t:7.1f} us {fl/us_t/1e6:7.1f} TF/s")

# fixed cost vs per-k-step cost of the N = 512 projections: sweep K
print("K sweep, N = 512 (us):  K  store_bf16  resid_f32")
for K in (64, 128, 256, 512, 1024, 2048):
    a, w = rn(M, K), rn(512, K)
    bias = torch.randn(512, generator=g).to(dev)
    o16 = torch.empty((M, 512), dtype=torch.bfloat16, device=dev)
    o32 = torch.randn((M, 512), generator=g).to(dev)
    t0 = timeit(lambda: dit_ops.gemm_bf16(a, w, bias, o16, 0))
    t3 = timeit(lambda: dit_ops.gemm_bf16(a, w, bias, o32, 3))
    print(f"  K={K:5d}  {t0:7.1f}  {t3:7.1f}")
