"""Is the wrong element of modulation_f32 under a concurrent gvf_gemm a LOST STORE (the output keeps what was there) or a wrong SUM?"""
import os, sys, threading, json, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvfdiffusion_amd.ops import dit_ops
from gvfdiffusion_amd import _lib

dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
g = torch.Generator().manual_seed(1)
C, N = 512, 56320
s2 = torch.randn((1, C), generator=g).to(dev)
W = (torch.randn((N, C), generator=g) * 0.05).to(dev)
b = torch.randn((N,), generator=g).to(dev)
a_ = torch.randn((32768, 512), generator=g).to(dev).half(); w_ = torch.randn((2048, 512), generator=g).to(dev).half()
o_ = torch.empty((32768, 2048), dtype=torch.float16, device=dev)
o32 = torch.empty((32768, 2048), dtype=torch.float32, device=dev)
ref = dit_ops.modulation_f32(s2, W, b).clone()
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
L = _lib.lib()

def mod_into(out):
    _lib.check(L.gvf_dit_modulation_f32(ctypes.c_void_p(s2.data_ptr()), 1, C, ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(b.data_ptr()), N,
                                        ctypes.c_void_p(out.data_ptr()), _lib.current_stream(dev)), "mod")

for other in ("gemm16", "gemm32", "gemm_small", "mm_torch"):
    stop = threading.Event(); res = []
    def A():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(sa):
            for _ in range(300):
                out = torch.full((1, N), 777.0, device=dev)
                mod_into(out)
                res.append(out)
            sa.synchronize()
        stop.set()
    def B():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(sb):
            while not stop.is_set():
                if other == "gemm16": dit_ops.gemm(a_, w_, None, o_, dit_ops.EPI_STORE_16)
                elif other == "gemm32": dit_ops.gemm(a_, w_, None, o32, dit_ops.EPI_STORE_F32)
                elif other == "gemm_small": dit_ops.gemm(a_[:2048], w_[:512], None, o_[:2048], dit_ops.EPI_STORE_16, n=512)
                elif other == "mm_torch": torch.mm(a_, w_.t())
                sb.synchronize()
    th = [threading.Thread(target=A), threading.Thread(target=B)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    n_bad = n_sentinel = 0; ex = []
    for r in res:
        d = (r != ref).reshape(-1).nonzero().reshape(-1)
        if d.numel():
            n_bad += 1
            vals = r.reshape(-1)[d]
            n_sentinel += int((vals == 777.0).sum())
            if len(ex) < 5: ex.append((d[:3].tolist(), [round(v, 4) for v in vals[:3].tolist()], [round(v, 4) for v in ref.reshape(-1)[d[:3]].tolist()]))
    print(json.dumps({"other stream": other, "divergent_launches": n_bad, "elements_left_at_sentinel": n_sentinel, "examples (idx, got, ref)": ex}), flush=True)
