"""End to end (BASELINE.json configs[4] at reduced sizes): DPM-Solver++ over the HIP DiT with two-scale CFG
-> de-normalise -> motion-VAE decode -> batched 4D render, i.e. inference_dpm_latent.py:225-269 of the
reference.  The same chain is run with the torch oracles as the model (oracle/dit_ref.py, oracle/vae_ref.py;
the sampler host code is shared) and the predicted deltas compared; the frames rendered from both delta sets
by the HIP rasteriser (itself pinned against oracle/rast_oracle.c in test_rast_gpu.py) must agree.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

TOL_DELTA_REL_L2 = 3e-2      # bf16 DiT x 8 model calls x bf16 VAE, vs the bf16-placement oracle chain
TOL_FRAME_PSNR_DB = 40.0     # frames rendered from the two delta sets


def test_sample_decode_render(cuda):
    from gvfdiffusion_amd import synthetic
    from gvfdiffusion_amd.model.autoencoder import GSKLTemporalVariationalAutoEncoder
    from gvfdiffusion_amd.model.dit import DiT
    from gvfdiffusion_amd.model.dpmsolver import DPM_Solver, NoiseScheduleVP, model_wrapper
    from gvfdiffusion_amd.model.gaussian_diffusion import create_gaussian_diffusion
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from oracle import dit_ref, vae_ref

    g = np.load(os.path.join(GOLD, "dit_small_golden.npz"))
    cfg = json.loads(bytes(g["cfg_json"]).decode())
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    B, T, N, C = 1, 3, 40, 16
    cond = dict(cond_images=torch.from_numpy(g["cond_images"][:B]), static_latent=torch.from_numpy(g["static_latent"][:B]),
                deformation_position_xyz=torch.from_numpy(g["xyz"][:B]))
    uncond = dict(cond, cond_images=torch.zeros_like(cond["cond_images"]))

    vcfg = dict(depth=2, dim=192, queries_dim=192, output_dim=14, num_inputs=64, num_latents=N, latent_dim=C, heads=3,
                dim_head=-1, num_timesteps=T, chunk_size=100)
    torch.manual_seed(3)
    vae = GSKLTemporalVariationalAutoEncoder(**vcfg)
    with torch.no_grad():
        for p in vae.parameters():
            p.copy_(torch.randn_like(p) * (1.0 / math.sqrt(p.shape[1]) if p.dim() == 2 else 0.05))
        vae.to_outputs.weight.mul_(0.05)         # deltas of a few % of the object size, as the trained decoder produces
    vsd = {k: v.detach().clone() for k, v in vae.state_dict().items()}

    P = 3000
    attrs = synthetic.random_gaussians(P, sh_degree=0, seed=4)
    gm = synthetic.gaussian_model_from(attrs, 0, torch.device("cpu"))
    # the 14 raw channels of the static Gaussians, as the reference feeds them to vae.decode (padded_static_gs)
    queries = torch.cat([gm._xyz, gm._features_dc.reshape(P, 3), gm._scaling, gm._rotation, gm._opacity], 1)[None].float()
    assert queries.shape == (1, P, 14)
    noise = torch.randn(B, T, N, C, generator=torch.Generator().manual_seed(5))
    mean, std = 0.02, 1.5

    diffusion = create_gaussian_diffusion(steps=1000, noise_schedule="linear", predict_type="v")
    ns = NoiseScheduleVP(schedule="discrete", betas=torch.tensor(diffusion.betas))

    def run_chain(model, decode, dev):
        c = {k: v.to(dev) for k, v in cond.items()}
        u = {k: v.to(dev) for k, v in uncond.items()}
        fn = model_wrapper(model, ns, model_type="v", model_kwargs={}, guidance_type="classifier-free", guidance_scale=3.0,
                           guidance_scale2=1.5, condition=c, unconditional_condition=u)
        x = DPM_Solver(fn, ns, algorithm_type="dpmsolver++").sample(x=noise.to(dev), steps=4, t_start=1.0, t_end=1 / 1000, order=2,
                                                                     skip_type="time_uniform", method="multistep")
        lat = (x * std + mean).reshape(B * T, N, C)
        return decode(lat, queries.to(dev)).float()

    dit = DiT(**cfg)
    dit.load_state_dict(sd, strict=True)
    dit = dit.to(cuda).eval()
    vae = vae.to(cuda)
    with torch.no_grad():
        d_hip = run_chain(dit, vae.decode, cuda)
        d_ref = run_chain(lambda x, t, **kw: dit_ref.dit_forward(sd, cfg, x, t, kw["cond_images"], kw["static_latent"],
                                                                 kw["deformation_position_xyz"], precision="bf16"),
                          lambda lat, q: vae_ref.vae_decode(vsd, vcfg, lat, q, T, "bf16"), torch.device("cpu"))
    assert d_hip.shape == (B, T, P, 14) and torch.isfinite(d_hip).all()
    rel = float((d_hip.cpu() - d_ref).norm() / d_ref.norm())
    assert rel < TOL_DELTA_REL_L2, rel

    gm = synthetic.gaussian_model_from(attrs, 0, cuda)
    rend = GaussianRenderer({"resolution": 128, "near": synthetic.NEAR, "far": synthetic.FAR, "ssaa": 1, "bg_color": (1, 1, 1)})
    rend.pipe.kernel_size = synthetic.KERNEL_2D
    cams = [synthetic.orbit_w2c(30.0 * i, 15.0) for i in range(4)]
    ext = torch.stack([cams[c] for t in range(T) for c in range(4)]).to(cuda)
    idx = [t for t in range(T) for _ in range(4)]
    K = synthetic.intrinsics().to(cuda)
    f_hip = rend.render_frames(gm, ext, K, delta_pc=d_hip[0].contiguous(), delta_index=idx).rgb
    f_ref = rend.render_frames(gm, ext, K, delta_pc=d_ref[0].to(cuda).contiguous(), delta_index=idx).rgb
    assert f_hip.shape == (T * 4, 3, 128, 128) and torch.isfinite(f_hip).all()
    assert float((f_hip < 0.99).float().mean()) > 0.02            # the object is in view
    assert float((f_hip[0] - f_hip[8]).abs().max()) > 1e-3        # and it moves between frames 0 and 2
    mse = float(((f_hip - f_ref) ** 2).mean())
    psnr = 10 * math.log10(1.0 / max(mse, 1e-20))
    print(f"pipeline: delta rel-L2 {rel:.2e}, frame PSNR {psnr:.1f} dB")
    assert psnr > TOL_FRAME_PSNR_DB, psnr
