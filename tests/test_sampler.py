"""Sampler restatement (gvfdiffusion_amd/model/dpmsolver.py, host logic on torch tensors) against
trajectories produced by the reference's model/dpmsolver.py (tests/golden/sampler_golden.npz)."""
import os

import numpy as np
import pytest
import torch

from gvfdiffusion_amd.model.dpmsolver import NoiseScheduleVP, model_wrapper, DPM_Solver, interpolate_fn
from gvfdiffusion_amd.model.gaussian_diffusion import create_gaussian_diffusion

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "sampler_golden.npz"))
CFG = dict(steps=1000, learn_sigma=False, sigma_small=False, use_kl=False, noise_schedule="cosine", predict_type="v",
           predict_xstart=False, rescale_timesteps=True, rescale_learned_sigmas=True)   # configs/diffusion.yml:16-25


def schedule():
    d = create_gaussian_diffusion(**CFG)
    return d, NoiseScheduleVP("discrete", betas=torch.from_numpy(d.betas))


def test_betas_and_schedule_tables():
    d, ns = schedule()
    assert np.array_equal(d.betas, G["betas"])                      # float64, bit-exact
    assert ns.total_N == int(G["total_N"]) == 996                   # cosine clip at lambda = -5.1
    assert np.array_equal(ns.log_alpha_array.numpy(), G["log_alpha_array"])
    assert np.array_equal(ns.t_array.numpy(), G["t_array"])
    t = torch.from_numpy(G["t_query"]).float()
    np.testing.assert_allclose(ns.marginal_alpha(t).numpy(), G["alpha_t"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(ns.marginal_std(t).numpy(), G["sigma_t"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(ns.marginal_lambda(t).numpy(), G["lambda_t"], rtol=2e-5, atol=2e-5)
    lam = torch.from_numpy(G["lam_query"]).float()
    np.testing.assert_allclose(ns.inverse_lambda(lam).numpy(), G["inv_lambda"], rtol=1e-5, atol=1e-6)
    # SURVEY appendix A known answers
    one = ns.marginal_alpha(torch.tensor([1.0])).item()
    assert abs(one - 0.00623376) < 1e-7 and abs(ns.inverse_lambda(torch.tensor([0.0])).item() - 0.49804196) < 1e-6


def test_interpolate_fn_extrapolates_with_outer_segments():
    xp = torch.tensor([[0.0, 1.0, 3.0]]); yp = torch.tensor([[0.0, 2.0, 3.0]])
    x = torch.tensor([[-1.0], [0.0], [0.5], [1.0], [2.0], [3.0], [5.0]])
    assert torch.allclose(interpolate_fn(x, xp, yp).reshape(-1), torch.tensor([-2.0, 0.0, 1.0, 2.0, 2.5, 3.0, 4.0]))


def toy_model(counter):
    def toy(x, t_input, cond_images=None, static_latent=None, deformation_position_xyz=None):
        counter["n"] += 1
        c = 0.0
        if cond_images is not None:
            c = c + 0.05 * cond_images.mean(dim=(1, 2, 3)).reshape(-1, 1, 1, 1)
        if static_latent is not None:
            c = c + 0.03 * static_latent.mean(dim=(1, 2)).reshape(-1, 1, 1, 1)
        return 0.3 * x * torch.cos(t_input / 200.0).reshape(-1, 1, 1, 1) + 0.1 * torch.sin(3 * x) + c
    return toy


@pytest.mark.parametrize("tag,scales", [("g11", (1.0, 1.0)), ("g23", (2.0, 3.0))])
def test_trajectories_match_reference(tag, scales):
    _, ns = schedule()
    xT = torch.from_numpy(G["xT"])
    cond = {"cond_images": torch.from_numpy(G["cond_images"]), "static_latent": torch.from_numpy(G["static_latent"]),
            "deformation_position_xyz": torch.from_numpy(G["xyz"])}
    uncond = dict(cond); uncond["cond_images"] = torch.zeros_like(cond["cond_images"])
    cnt = {"n": 0}
    mf = model_wrapper(toy_model(cnt), ns, model_type="v", model_kwargs={}, guidance_type="classifier-free",
                       guidance_scale=scales[0], guidance_scale2=scales[1], condition=cond, unconditional_condition=uncond)
    w = mf(xT, torch.tensor([0.7, 0.7]))
    np.testing.assert_allclose(w.numpy(), G[f"wrap_{tag}"], rtol=1e-5, atol=1e-6)
    solver = DPM_Solver(mf, ns, algorithm_type="dpmsolver++")
    for steps in (4, 20, 32):
        cnt["n"] = 0
        out = solver.sample(xT, steps=steps, t_start=1.0, t_end=1 / 1000, order=2, skip_type="time_uniform",
                            method="multistep")
        assert cnt["n"] == int(G[f"multistep_{tag}_{steps}_nfe"]) == steps       # exactly `steps` NFEs
        np.testing.assert_allclose(out.numpy(), G[f"multistep_{tag}_{steps}"], rtol=2e-4, atol=2e-5)
    cnt["n"] = 0
    out = solver.sample(xT, steps=12, t_start=1.0, t_end=1 / 1000, order=2, skip_type="time_uniform", method="singlestep")
    assert cnt["n"] == int(G[f"singlestep_{tag}_12_nfe"])
    np.testing.assert_allclose(out.numpy(), G[f"singlestep_{tag}_12"], rtol=2e-4, atol=2e-5)
    cnt["n"] = 0
    out = solver.sample(xT, steps=100, t_start=1.0, t_end=1 / 1000, order=2, skip_type="time_uniform", method="adaptive")
    # same accept / reject decisions: the NFE the solver REPORTS is the reference's count (order evaluations per attempted step); it runs fewer,
    # because a rejected step keeps its model(x, s) for the retry (same x, same s: the reference evaluates it again)
    nfe_ref = int(G[f"adaptive_{tag}_nfe"])
    assert solver.last_nfe == nfe_ref and cnt["n"] == nfe_ref - solver.spec_stats["rejected"]
    solver.speculate = True                                                         # (CPU tensors never speculate: the switch changes nothing here)
    cnt["n"] = 0
    out2 = solver.sample(xT, steps=100, t_start=1.0, t_end=1 / 1000, order=2, skip_type="time_uniform", method="adaptive")
    assert torch.equal(out, out2) and cnt["n"] == nfe_ref - solver.spec_stats["rejected"]
    np.testing.assert_allclose(out.numpy(), G[f"adaptive_{tag}"], rtol=5e-4, atol=5e-5)


def test_every_solver_branch_matches_reference():
    """algorithm x method x order x solver_type x skip_type, 13 NFEs each, against the reference's output."""
    _, ns = schedule()
    cnt = {"n": 0}
    mf = model_wrapper(toy_model(cnt), ns, model_type="v", guidance_type="uncond")
    xT = torch.from_numpy(G["xT"])
    keys = [k for k in G.files if k.startswith("var_")]
    assert len(keys) == 28
    for key in keys:
        _, alg, rest = key.split("_", 2)
        method = "singlestep_fixed" if rest.startswith("singlestep_fixed") else rest.split("_")[0]
        order, skip_a, skip_b, st = rest[len(method) + 1:].split("_") if "time" in rest else (None,) * 4
        if order is None:
            order, skip, st = rest[len(method) + 1:].split("_")
        else:
            skip = f"{skip_a}_{skip_b}"
        out = DPM_Solver(mf, ns, algorithm_type=alg).sample(xT, steps=13, t_start=1.0, t_end=1 / 1000, order=int(order),
                                                            skip_type=skip, method=method, solver_type=st,
                                                            denoise_to_zero=(st == "taylor"))
        np.testing.assert_allclose(out.numpy(), G[key], rtol=5e-4, atol=5e-5, err_msg=key)
    x, inter = DPM_Solver(mf, ns).sample(xT, steps=6, order=2, method="multistep", return_intermediate=True, denoise_to_zero=True)
    assert len(inter) == 8 and torch.equal(inter[-1], x)
