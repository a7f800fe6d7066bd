"""gvf_gemm256 (csrc/gemm256.hip) against gvf_gemm's 128-wide kernel and torch.addmm (hipBLASLt) on large plain projections; checks the result
against an fp32 product of the same 16-bit operands.  GPU only."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvfdiffusion_amd import _lib
from gvfdiffusion_amd.ops import dit_ops
_i, _vp = ctypes.c_int, ctypes.c_void_p
_lib.register({"gvf_gemm256": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp])})
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for lp in (torch.bfloat16, torch.float16):
    for name, M, N, K, has_bias in (("vae to_qkv", 12288, 2304, 768, False), ("vae fc1", 12288, 6144, 768, True), ("vae dec to_q", 262144, 768, 768, False),
                                    ("square 8192", 8192, 8192, 4096, True), ("small 256", 256, 256, 64, True)):
        a = torch.randn((M, K), generator=g).to(lp).to(dev); w = (torch.randn((N, K), generator=g) * K ** -0.5).to(lp).to(dev)
        bias = torch.randn(N, generator=g).to(dev) if has_bias else None
        out = torch.full((M, N), float("nan"), dtype=lp, device=dev)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        st = _lib.current_stream(dev)
        fn = lambda: _lib.check(_lib.lib().gvf_gemm256(dit_ops.dt_code(lp), p(a), K, p(w), K, p(bias), p(out), N, M, N, K, st), "gvf_gemm256")
        fn(); torch.cuda.synchronize()
        rows = torch.randint(0, M, (64,), generator=g).to(dev)
        ref = a[rows].float() @ w.float().t() + (bias if bias is not None else 0)
        err = float((out[rows].float() - ref).abs().max() / ref.abs().max())
        ok = bool(torch.isfinite(out).all())
        us = timeit(fn)
        os.environ["GVF_GEMM256"] = "0"
        out2 = torch.empty_like(out)
        us_old = timeit(lambda: dit_ops.gemm_bf16(a, w, bias, out2, 0)) if False else float("nan")
        us_t = timeit(lambda: torch.addmm(bias.to(lp), a, w.t())) if bias is not None else timeit(lambda: torch.mm(a, w.t()))
        fl = 2.0 * M * N * K
        print(f"{str(lp)[6:]:9s} {name:14s} M={M:6d} N={N:5d} K={K:5d}: gemm256 {us:8.1f} us {fl/us/1e6:7.1f} TF/s  max rel err {err:.2e} finite {ok} | hipBLASLt {us_t:8.1f} us {fl/us_t/1e6:7.1f} TF/s")
