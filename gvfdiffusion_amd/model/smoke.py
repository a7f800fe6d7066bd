"""DiT leg of __graft_entry__.smoke(): one small denoise step on cuda:0, checked against the torch oracle."""
import json
import os

import numpy as np
import torch


def run(dev):
    from gvfdiffusion_amd.model.dit import DiT
    from oracle import dit_ref                       # checker only (smoke is allowed to use the oracle)
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    g = np.load(os.path.join(root, "tests", "golden", "dit_small_golden.npz"))
    cfg = json.loads(bytes(g["cfg_json"]).decode())
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    model = DiT(**cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    args = [torch.from_numpy(g[k]).to(dev) for k in ("x", "t", "cond_images", "static_latent", "xyz")]
    y = model(*args)
    ref = dit_ref.dit_forward({k: v.to(dev) for k, v in sd.items()}, cfg, *args, precision="bf16")
    gold = torch.from_numpy(g["y"]).to(dev)
    rb = float((y - ref).norm() / ref.norm())
    rg = float((y - gold).norm() / gold.norm())
    assert rb < 1e-2 and rg < 3e-2, (rb, rg)
    print(f"smoke: DiT denoise step {tuple(y.shape)}: rel_l2 vs bf16 oracle {rb:.2e}, vs fp32 reference golden {rg:.2e} OK")
