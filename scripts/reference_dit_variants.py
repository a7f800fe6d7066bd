"""What the reference's two un-built DiT options do when switched on (runs in the BUILD container only: imports /root/reference).

    python scripts/reference_dit_variants.py

`share_mod=True` (model/dit.py:161, 324) and `pe_mode='rope'` (model/dit.py:320, 376) are off in configs/diffusion.yml; gvfdiffusion_amd's
DiT refuses both at construction.  This script records, with the reference's own classes on CPU, whether the reference itself can run them --
the output is quoted in DESIGN.md section 3.1."""
import importlib.util
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
mg.install_stubs()
sys.path.insert(0, mg.REF)
from model.dit import DiT  # noqa: E402


def run(tag, **over):
    cfg = dict(mg.DIT_SMALL, **over)
    print(f"== {tag}: {over}")
    try:
        torch.manual_seed(0)
        model = DiT(**cfg).eval()
        mg._randomise(model, 1)
        g = torch.Generator().manual_seed(2)
        B, T, N = 2, 3, 40
        x = torch.randn((B, T, N, 16), generator=g)
        t = torch.tensor([998.996, 431.25])
        cond = torch.randn((B, T, 37, 32), generator=g)
        static = torch.randn((B, 50, 14), generator=g)
        xyz = torch.rand((B, N, 3), generator=g) - 0.5
        with torch.no_grad():
            y = model(x, t, cond_images=cond, static_latent=static, deformation_position_xyz=xyz)
        print("   forward ran:", tuple(y.shape), "mean |y| = %.4f" % float(y.abs().mean()))
    except Exception as e:                     # noqa: BLE001
        tb = traceback.extract_tb(e.__traceback__)
        where = [f for f in tb if "/root/reference" in f.filename]
        at = where[-1] if where else tb[-1]
        print(f"   FAILS: {type(e).__name__}: {str(e).splitlines()[0][:200]}")
        print(f"   at {at.filename.replace('/root/reference/', '')}:{at.lineno}: {at.line}")


run("baseline (the fixture's configuration)")
run("share_mod, temporal attention on", share_mod=True)
run("share_mod, temporal attention off", share_mod=True, no_temporal_attn=True)
run("rope", pe_mode="rope")
run("rope without q/k RMS norm", pe_mode="rope", qk_rms_norm=False)
