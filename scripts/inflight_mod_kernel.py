"""modulation_f32 (the fp32 adaLN GEMV of the DiT step) alone on one stream while gvf_gemm runs on another: are its results bitwise stable?"""
import os, sys, threading, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvfdiffusion_amd.ops import dit_ops

dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
g = torch.Generator().manual_seed(1)
C, N = 512, 56320
s2 = torch.randn((1, C), generator=g).to(dev)
W = (torch.randn((N, C), generator=g) * 0.05).to(dev)
b = torch.randn((N,), generator=g).to(dev)
a_ = torch.randn((32768, 512), generator=g).to(dev).half(); w_ = torch.randn((2048, 512), generator=g).to(dev).half()
o_ = torch.empty((32768, 2048), dtype=torch.float16, device=dev)
ref = dit_ops.modulation_f32(s2, W, b).clone()
ref_t = (s2 @ W.t() + b)
print("vs torch:", float((ref - ref_t).abs().max()))
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for other in ("nothing", "gemm", "fill"):
    stop = threading.Event(); res = []
    def A():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(sa):
            for _ in range(300):
                res.append(dit_ops.modulation_f32(s2, W, b))
            sa.synchronize()
        stop.set()
    def B():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(sb):
            while not stop.is_set():
                if other == "gemm":
                    dit_ops.gemm(a_, w_, None, o_, dit_ops.EPI_STORE_16)
                elif other == "fill":
                    o_.fill_(1.0)
                sb.synchronize()
    th = [threading.Thread(target=A), threading.Thread(target=B)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    nbad, ex = 0, None
    for r in res:
        d = (r != ref)
        if bool(d.any()):
            nbad += 1
            if ex is None:
                idx = d.reshape(-1).nonzero().reshape(-1)
                ex = {"n_elements": int(idx.numel()), "first_indices": idx[:12].tolist(), "got": r.reshape(-1)[idx[:4]].tolist(), "ref": ref.reshape(-1)[idx[:4]].tolist()}
    print(json.dumps({"other stream": other, "launches": len(res), "divergent_launches": nbad, "example": ex}), flush=True)

# ---- which elements, and are they stray WRITES into the output or wrong SUMS?  (a) every divergent launch's indices; (b) zero-filled tensors
# allocated on stream A while the GEMM runs on B must stay zero; (c) the same GEMV as torch.mv (library kernel) under the same conditions
for mode in ("indices", "zeros", "torch_mv"):
    stop = threading.Event(); res = []
    def A2():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(sa):
            for _ in range(300):
                if mode == "zeros":
                    res.append(torch.zeros((1, N), device=dev))
                elif mode == "torch_mv":
                    res.append(torch.addmv(b, W, s2[0])[None])
                else:
                    res.append(dit_ops.modulation_f32(s2, W, b))
            sa.synchronize()
        stop.set()
    def B2():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(sb):
            while not stop.is_set():
                dit_ops.gemm(a_, w_, None, o_, dit_ops.EPI_STORE_16)
                sb.synchronize()
    th = [threading.Thread(target=A2), threading.Thread(target=B2)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    r0 = torch.zeros((1, N), device=dev) if mode == "zeros" else (res[0] if mode == "torch_mv" else ref)
    info = []
    for j, r in enumerate(res):
        d = (r != r0)
        if bool(d.any()):
            idx = d.reshape(-1).nonzero().reshape(-1)
            info.append((j, idx[:4].tolist(), [round(v, 5) for v in (r.reshape(-1)[idx[:2]] - r0.reshape(-1)[idx[:2]]).tolist()]))
    print(json.dumps({"mode": mode, "divergent": len(info), "first": info[:10]}), flush=True)
print("W intact:", bool(torch.equal(dit_ops.modulation_f32(s2, W, b), ref)))
