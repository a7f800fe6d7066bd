"""Times the attention kernel on the DiT's shapes (B=1, T=24, N=512, H=16, d=32) and the VAE's (d=64); GPU only."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvfdiffusion_amd.ops import dit_ops

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
def rn(*s): return torch.randn(s, generator=g).to(torch.bfloat16).to(dev)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

T, N, H, C = 24, 512, 16, 512
M = T * N
res = []
# static cross: K/V shared by the frames
for name, nb_kv, Lk in (("static cross Lk4096", 1, 4096), ("image cross Lk1370", T, 1370)):
    q = rn(M, C); out = torch.empty_like(q)
    kc = rn(nb_kv, H, Lk, 32)
    Lp = (Lk + 63) // 64 * 64
    vt = torch.zeros((nb_kv, H, 32, Lp), dtype=torch.bfloat16, device=dev); vt[..., :Lk] = rn(nb_kv, H, 32, Lk)
    if nb_kv == 1:
        fn = lambda: dit_ops.attention_bf16(q, kc, vt, out, 1, T, N, Lk, H, (T * N * C, N * C, C), (H * Lk * 32, 0, 32, Lk * 32), (H * 32 * Lp, 0, Lp, 32 * Lp), (T * N * C, N * C, C), v_transposed=True)
    else:
        fn = lambda: dit_ops.attention_bf16(q, kc, vt, out, T, 1, N, Lk, H, (N * C, 0, C), (H * Lk * 32, 0, 32, Lk * 32), (H * 32 * Lp, 0, Lp, 32 * Lp), (N * C, 0, C), v_transposed=True)
    us = timeit(fn); fl = 4.0 * M * Lk * C
    res.append((name, us, fl / us / 1e6))
qkv = rn(M, 3 * C); ab = torch.empty((M, C), dtype=torch.bfloat16, device=dev)
gq = torch.ones((H, 32), device=dev); gk = torch.ones((H, 32), device=dev)
s3 = (N * 3 * C, 0, 3 * C)
us = timeit(lambda: dit_ops.attention_bf16(qkv, qkv[:, C:], qkv[:, 2 * C:], ab, T, 1, N, N, H, s3, s3, s3, (N * C, 0, C), gq, gk))
res.append(("spatial self L512 rms", us, 4.0 * M * N * C / us / 1e6))
# VAE decoder cross attention: 262144 queries x 512 latents, 12 heads of 64
P, Hv, Lk = 262144, 12, 512
q = rn(P, Hv * 64); k = rn(Lk, Hv * 64); v = rn(Lk, Hv * 64); o = torch.empty_like(q)
us = timeit(lambda: dit_ops.attention_bf16(q, k, v, o, 1, 1, P, Lk, Hv, (0, 0, Hv * 64, 64), (0, 0, Hv * 64, 64), (0, 0, Hv * 64, 64), (0, 0, Hv * 64, 64), head_dim=64), n=5)
res.append(("vae d64 Lq262144 Lk512", us, 4.0 * P * Lk * Hv * 64 / us / 1e6))
for name, us, tf in res:
    print(f"{name:28s} {us:9.1f} us  {tf:7.1f} TFLOP/s  ({tf / 2500 * 100:.1f}% of bf16 MFMA peak)")
