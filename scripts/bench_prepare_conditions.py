"""DiT.prepare_conditions at configs/diffusion.yml, B = 1, T = 24 (1370 image tokens per frame, 4096 static tokens): the step-invariant condition
projections + every block's to_kv(context) + the tiled K / V caches -- per-sample cost outside the denoise step.  Prints ms per call for
both operand types (fp16 also orders the keys of every cache by norm) and lists the kernels a call launches."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gvfdiffusion_amd import synthetic                       # noqa: E402
from gvfdiffusion_amd.model.dit import DiT                    # noqa: E402

dev = torch.device("cuda:0")
man = json.load(open(os.path.join(ROOT, "tests", "golden", "dit_manifest.json")))
model = DiT(**man["config"])
model.load_state_dict(synthetic.dit_state_dict(man["state_dict"], seed=0), strict=True)
model = model.to(dev).eval()
inp = {k: v.to(dev) for k, v in synthetic.dit_inputs(B=1, T=24, seed=1).items()}
for name in ("fp16", "bf16"):
    model.set_compute_dtype(name)
    for rep in range(6):
        model.invalidate_conditions()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.prepare_conditions(inp["cond_images"], inp["static_latent"], inp["deformation_position_xyz"], 24)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep >= 2:
            print(f"prepare_conditions [{name}] call {rep}: {dt * 1e3:.2f} ms")
