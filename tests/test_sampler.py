"""Sampler restatement (gvfdiffusion_amd/model/dpmsolver.py, host logic on torch tensors) against
trajectories produced by the reference's model/dpmsolver.py (tests/golden/sampler_golden.npz).

The two trajectory tests run twice: with the state on the CPU (the `-m "not gpu"` suite) and with the state on the MI355X (`gpu`-marked
parametrisation: the driver's GPU run then holds the product's solver against the REFERENCE's trajectories too -- the GPU chain tests of
test_pipeline_gpu.py / test_dit_gpu.py have the product's solver on both sides)."""
import math
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


def test_scalar_schedule_fast_path_is_bitwise_the_tensor_path():
    """NoiseScheduleVP answers one-element host queries (what every solver in this package asks) with numpy float32 scalar arithmetic instead
    of interpolate_fn's ~15 tensor operations (model/dpmsolver.py:1270-1309 restated): same formula, same correctly rounded binary32 operations,
    so the SAME BITS -- inside the table, on its knots, and beyond both ends (the outer segments extrapolate)."""
    import random
    _, ns = schedule()

    def slow_la(t):
        return interpolate_fn(t.reshape((-1, 1)).to(ns.t_array.dtype), ns.t_array, ns.log_alpha_array).reshape((-1))

    def slow_inv(lam):
        la = -0.5 * torch.logaddexp(torch.zeros((1,)), -2.0 * lam)
        return interpolate_fn(la.reshape((-1, 1)), ns._la_flip, ns._t_flip).reshape((-1,))
    random.seed(0)
    for _ in range(3000):
        t = torch.tensor([random.choice([random.uniform(-0.1, 1.1), random.uniform(0, 0.01), float(ns.t_array[0, random.randrange(ns.total_N)])])])
        assert torch.equal(ns.marginal_log_mean_coeff(t), slow_la(t)), t
        lam = torch.tensor([random.uniform(-8, 12)])
        assert torch.equal(ns.inverse_lambda(lam), slow_inv(lam)), lam
    t64 = torch.tensor([0.3], dtype=torch.float64)                      # a float64 query is answered in the table's precision, as before
    assert ns.marginal_log_mean_coeff(t64).dtype == torch.float64 and float(ns.marginal_log_mean_coeff(t64)) == float(slow_la(t64))
    many = torch.tensor([0.2, 0.7])                                     # more than one element: the tensor path
    assert torch.equal(ns.marginal_log_mean_coeff(many), slow_la(many))


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


DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def _device(name):
    if name == "cuda" and not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device(name)


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("tag,scales", [("g11", (1.0, 1.0)), ("g23", (2.0, 3.0))])
def test_trajectories_match_reference(tag, scales, dev):
    dev = _device(dev)
    _, ns = schedule()
    xT = torch.from_numpy(G["xT"]).to(dev)
    cond = {"cond_images": torch.from_numpy(G["cond_images"]).to(dev), "static_latent": torch.from_numpy(G["static_latent"]).to(dev),
            "deformation_position_xyz": torch.from_numpy(G["xyz"]).to(dev)}
    uncond = dict(cond); uncond["cond_images"] = torch.zeros_like(cond["cond_images"])
    cnt = {"n": 0}
    mf = model_wrapper(toy_model(cnt), ns, model_type="v", model_kwargs={}, guidance_type="classifier-free",
                       guidance_scale=scales[0], guidance_scale2=scales[1], condition=cond, unconditional_condition=uncond)
    w = mf(xT, torch.tensor([0.7, 0.7]))
    np.testing.assert_allclose(w.cpu().numpy(), G[f"wrap_{tag}"], rtol=1e-5, atol=1e-6)
    solver = DPM_Solver(mf, ns, algorithm_type="dpmsolver++")
    for steps in (4, 20, 32):
        cnt["n"] = 0
        out = solver.sample(xT, steps=steps, t_start=1.0, t_end=1 / 1000, order=2, skip_type="time_uniform",
                            method="multistep")
        assert cnt["n"] == int(G[f"multistep_{tag}_{steps}_nfe"]) == steps       # exactly `steps` NFEs
        assert out.device.type == dev.type
        np.testing.assert_allclose(out.cpu().numpy(), G[f"multistep_{tag}_{steps}"], rtol=2e-4, atol=2e-5)
    cnt["n"] = 0
    out = solver.sample(xT, steps=12, t_start=1.0, t_end=1 / 1000, order=2, skip_type="time_uniform", method="singlestep")
    assert cnt["n"] == int(G[f"singlestep_{tag}_12_nfe"])
    np.testing.assert_allclose(out.cpu().numpy(), G[f"singlestep_{tag}_12"], rtol=2e-4, atol=2e-5)
    cnt["n"] = 0
    out = solver.sample(xT, steps=100, t_start=1.0, t_end=1 / 1000, order=2, skip_type="time_uniform", method="adaptive")
    # same accept / reject decisions: the NFE the solver REPORTS is the reference's count (order evaluations per attempted step); it runs fewer,
    # because a rejected step keeps its model(x, s) for the retry (same x, same s: the reference evaluates it again)
    nfe_ref = int(G[f"adaptive_{tag}_nfe"])
    assert solver.last_nfe == nfe_ref and cnt["n"] == nfe_ref - solver.spec_stats["rejected"]
    solver.speculate = True                                    # (CPU tensors never speculate: the switch changes nothing there)
    cnt["n"] = 0
    out2 = solver.sample(xT, steps=100, t_start=1.0, t_end=1 / 1000, order=2, skip_type="time_uniform", method="adaptive")
    st = solver.spec_stats
    assert torch.equal(out, out2) and solver.last_nfe == nfe_ref
    if dev.type == "cpu":
        assert cnt["n"] == nfe_ref - st["rejected"] and st["speculated"] == 0
    else:
        # on the device the next step's first evaluation is queued before the host waits for the error norm: one evaluation per attempted step
        # is speculated (not behind the last), a rejected step drops its own; accepted steps then start from the speculated evaluation
        assert st["speculated"] >= st["steps"] - 1 - st["rejected"] and st["dropped"] <= st["rejected"]
        assert cnt["n"] == nfe_ref - st["rejected"] + st["dropped"]
    np.testing.assert_allclose(out.cpu().numpy(), G[f"adaptive_{tag}"], rtol=5e-4, atol=5e-5)


@pytest.mark.parametrize("dev", DEVICES)
def test_every_solver_branch_matches_reference(dev):
    """algorithm x method x order x solver_type x skip_type, 13 NFEs each, against the reference's output."""
    dev = _device(dev)
    _, ns = schedule()
    cnt = {"n": 0}
    mf = model_wrapper(toy_model(cnt), ns, model_type="v", guidance_type="uncond")
    xT = torch.from_numpy(G["xT"]).to(dev)
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
        if dev.type == "cuda" and method == "adaptive":
            # The adaptive walks of this table start at t = 1 with h = 0.05, where the two orders agree to ROUNDING NOISE (E ~ 6e-7), and the
            # second step's size is 0.9 h E^(-1/order): whatever the last bits of that noise are on the machine at hand decides it (this
            # container: h2 = 5.24, the MI355X box's host AND device: 5.38; both printed by a DPM_Solver with .trace = []).  Every later decision
            # is far from its threshold and the walks end within the solver's own tolerance (atol 0.0078, rtol 0.05 per step) of each other:
            # measured 1.1e-2 max-abs on the order-3 noise-prediction variant, identical for the box's CPU and the device.  So on the device the
            # adaptive variants are held to that class of bar (whole-tensor rel-L2 < 0.1: two walks with rtol 0.05 per step); the bits-level bar against the reference's trajectory is the host parametrisation
            # (fixtures generated in this container) and test_trajectories_match_reference[cuda] (dpmsolver++ order 2: 5e-4 on the device too).
            # (order-3 Taylor variant with denoise_to_zero: 0.27 max-abs, 9 % on single elements, the same on the box's CPU.)
            ref = torch.from_numpy(G[key]).double()
            rel = float((out.cpu().double() - ref).norm() / ref.norm())
            print(f"{key} on the device: rel-L2 vs the reference's walk {rel:.2e}")
            assert rel < 0.1, (key, rel)
            continue
        np.testing.assert_allclose(out.cpu().numpy(), G[key], rtol=5e-4, atol=5e-5, err_msg=key)
    x, inter = DPM_Solver(mf, ns).sample(xT, steps=6, order=2, method="multistep", return_intermediate=True, denoise_to_zero=True)
    assert len(inter) == 8 and torch.equal(inter[-1], x)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 24, 512, 16), (3, 5, 7, 3), (2, 1, 1, 1), (1, 1023)])
def test_fused_solver_launches_are_the_reference_expressions_rounded_operation_by_operation(shape):
    """csrc/dpm.hip (gvf_dpm_x0 / gvf_dpm_lincomb / gvf_dpm_second_err): the DPM-Solver state updates as single launches.  Every product, sum and
    quotient is rounded to fp32 on its own in the order the reference's expressions evaluate (model/dpmsolver.py:450-461, 564-609, 611-690,
    1013-1019), which is what a chain of CPU tensor operations does too: the device results equal the CPU chain BIT FOR BIT (sizes with and
    without whole 16-byte pieces; three samples); the error norm against a float64 evaluation of the reference's formula."""
    from gvfdiffusion_amd.ops import dit_ops
    dev = _device("cuda")
    g = torch.Generator().manual_seed(sum(shape))
    x, m, m1, xp = (torch.randn(shape, generator=g) for _ in range(4))
    m1 = m + 0.05 * m1
    xp = x + 0.1 * xp
    a, b, c, sig, alp, atol, rtol = 0.83721, 0.41234, 0.27391, 0.6123, 0.7906, 0.0078, 0.05
    f = np.float32
    X, M, M1, XP = (t.to(dev) for t in (x, m, m1, xp))
    xn, mn, m1n, xpn = (t.numpy() for t in (x, m, m1, xp))          # numpy float32: one correctly rounded IEEE operation per operator, no reciprocal tricks
    assert dit_ops.dpm_fusable(X, M, M1, XP) and not dit_ops.dpm_fusable(X, M.double()) and not dit_ops.dpm_fusable(X, m)
    assert np.array_equal(dit_ops.dpm_x0(X, M, sig, alp).cpu().numpy(), (xn - f(sig) * mn) / f(alp))
    assert np.array_equal(dit_ops.dpm_lincomb(X, M, a, -b).cpu().numpy(), f(a) * xn + f(-b) * mn)
    assert np.array_equal(dit_ops.dpm_lincomb(X, M, a, -b, M1, c).cpu().numpy(), (f(a) * xn + f(-b) * mn) + f(c) * m1n)
    lo, hi, E = dit_ops.dpm_second_err(X, M, M1, XP, a, b, c, atol, rtol)
    lo_ref = f(a) * xn - f(b) * mn
    hi_ref = lo_ref - f(c) * (m1n - mn)
    assert lo_ref.dtype == np.float32 and np.array_equal(lo.cpu().numpy(), lo_ref) and np.array_equal(hi.cpu().numpy(), hi_ref)
    delta = np.maximum(f(atol), f(rtol) * np.maximum(np.abs(lo_ref), np.abs(xpn)))
    v = ((hi_ref - lo_ref) / delta).astype(np.float64).reshape(shape[0], -1)
    E_ref = float(np.sqrt((v * v).mean(axis=-1)).max())
    assert E.shape == (1,) and float(E) == pytest.approx(E_ref, rel=2e-6)
    first = float(E)
    for _ in range(3):                              # fixed summation order: the same bits every launch
        assert float(dit_ops.dpm_second_err(X, M, M1, XP, a, b, c, atol, rtol)[2]) == first
    XN = X.clone(); XN.view(-1)[0] = float("nan")
    assert math.isnan(float(dit_ops.dpm_second_err(XN, M, M1, XP, a, b, c, atol, rtol)[2]))


@pytest.mark.gpu
def test_fused_and_chained_solver_steps_walk_the_same_trajectories():
    """GVF_DPM_FUSED=0 switches the device solver back to the chains of tensor operations: the same walks -- multistep, singlestep and adaptive
    dpmsolver++ of order 2 -- end within rounding of each other, with the same accept / reject decisions."""
    dev = _device("cuda")
    _, ns = schedule()
    xT = torch.from_numpy(G["xT"]).to(dev)
    outs = {}
    for flag in ("1", "0"):
        os.environ["GVF_DPM_FUSED"] = flag
        try:
            cnt = {"n": 0}
            solver = DPM_Solver(model_wrapper(toy_model(cnt), ns, model_type="v", guidance_type="uncond"), ns, algorithm_type="dpmsolver++")
            res = [solver.sample(xT, steps=20, order=2, method="multistep"), solver.sample(xT, steps=12, order=2, method="singlestep")]
            solver.trace = []
            res.append(solver.sample(xT, steps=100, order=2, method="adaptive", t_start=0.9))
            outs[flag] = (res, solver.last_nfe, dict(solver.spec_stats), list(solver.trace))
        finally:
            os.environ.pop("GVF_DPM_FUSED", None)
    for a_, b_ in zip(outs["1"][0], outs["0"][0]):
        assert float((a_ - b_).abs().max()) < 2e-5
    assert outs["1"][1:3] == outs["0"][1:3]
    for (s1, t1, h1, e1), (s0, t0, h0, e0) in zip(outs["1"][3], outs["0"][3]):
        assert abs(s1 - s0) < 1e-5 and abs(t1 - t0) < 1e-5 and abs(e1 - e0) < 1e-3 * max(e0, 1e-3)
