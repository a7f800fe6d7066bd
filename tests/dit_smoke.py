"""DiT leg of __graft_entry__.smoke(): one small denoise step on cuda:0, checked against the torch oracle."""
import json
import os

import numpy as np
import torch


def run(dev):
    from gvfdiffusion_amd.model.dit import DiT
    from oracle import dit_ref                       # checker only (smoke is allowed to use the oracle)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dit_small_golden.npz"))
    cfg = json.loads(bytes(g["cfg_json"]).decode())
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    model = DiT(**cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    args = [torch.from_numpy(g[k]).to(dev) for k in ("x", "t", "cond_images", "static_latent", "xyz")]
    gold = torch.from_numpy(g["y"]).to(dev)
    sdc = {k: v.to(dev) for k, v in sd.items()}
    for name, lp, tol_o, tol_g in (("fp16", torch.float16, 1e-4, 2e-4), ("bf16", torch.bfloat16, 4e-4, 2e-3)):
        y = model.set_compute_dtype(lp)(*args)
        ref = dit_ref.dit_forward(sdc, cfg, *args, precision=name)
        ro = float((y - ref).norm() / ref.norm())
        rg = float((y - gold).norm() / gold.norm())
        assert ro < tol_o and rg < tol_g, (name, ro, rg)
        print(f"smoke: DiT denoise step {tuple(y.shape)} [{name}]: rel_l2 vs {name} oracle {ro:.2e}, vs fp32 reference golden {rg:.2e} OK")
