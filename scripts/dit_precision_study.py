"""CPU study (no GPU): which operand type at which rounding site of the DiT pipeline buys how much accuracy.  Runs the oracle
restatement (oracle/dit_ref.py) of the full configs/diffusion.yml forward with the pipeline's rounding points set per site class and
prints rel-L2 against the reference's fp32 golden (tests/golden/dit_full_golden.npz) next to the reference's own autocast errors
(tests/golden/dit_autocast_golden.npz).  ~20 s per forward on 8 threads."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dit_ref          # noqa: E402
from gvfdiffusion_amd import synthetic   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    man = json.load(open(os.path.join(GOLD, "dit_manifest.json")))
    sd = synthetic.dit_state_dict(man["state_dict"], seed=0)
    inp = synthetic.dit_inputs(B=1, T=24, seed=1)
    gold = torch.from_numpy(np.load(os.path.join(GOLD, "dit_full_golden.npz"))["y"])
    ac = np.load(os.path.join(GOLD, "dit_autocast_golden.npz"))
    print({k: float(ac[k]) for k in ac.files if ac[k].size == 1})
    cases = sys.argv[1:] or ["bf16", "fp16", "gemm=fp16,attn=bf16", "gemm=bf16,attn=fp16"]
    for c in cases:
        prec = dict(kv.split("=") for kv in c.split(",")) if "=" in c else c
        t0 = time.time()
        with torch.no_grad():
            y = dit_ref.dit_forward(sd, man["config"], inp["x"], inp["t"], inp["cond_images"], inp["static_latent"],
                                    inp["deformation_position_xyz"], precision=prec)
        print(f"{c:28s} rel_l2 vs fp32 golden {rel_l2(y, gold):.3e}   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
