"""Host-side contract of the DiT module (no GPU): constructor surface, state_dict compatibility with the
reference's 446-tensor checkpoint layout, loud failure on CPU tensors."""
import json
import os

import numpy as np
import pytest
import torch

from gvfdiffusion_amd.model.dit import DiT
from gvfdiffusion_amd import _lib

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_state_dict_matches_reference_manifest():
    man = json.load(open(os.path.join(GOLD, "dit_manifest.json")))
    torch.manual_seed(0)
    model = DiT(**man["config"])
    sd = model.state_dict()
    assert len(sd) == 446
    assert {k: list(v.shape) for k, v in sd.items()} == man["state_dict"]
    # reference initialisation invariants (model/dit.py:414-427): zero adaLN + zero output head, unit RMS gains
    assert float(sd["blocks.3.adaLN_modulation.1.weight"].abs().max()) == 0
    assert float(sd["final_layer.linear.weight"].abs().max()) == 0
    assert float(sd["blocks.0.adaLN_modulation_temporal.1.weight"].abs().max()) > 0
    assert torch.all(sd["blocks.0.spatial_self_attn.q_rms_norm.gamma"] == 1)
    assert "blocks.0.image_cross_attn.q_rms_norm.gamma" not in sd          # qk_rms_norm_cross = False
    n_params = sum(p.numel() for p in model.parameters())
    assert abs(n_params / 1e6 - 105.51) < 0.01


def test_small_golden_state_dict_loads_strictly():
    g = np.load(os.path.join(GOLD, "dit_small_golden.npz"))
    cfg = json.loads(bytes(g["cfg_json"]).decode())
    model = DiT(**cfg)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    model.load_state_dict(sd, strict=True)
    model.load_state_dict({"module." + k: v for k, v in sd.items()}, strict=False)   # prefixed keys are simply ignored


def test_forward_refuses_cpu_tensors():
    g = np.load(os.path.join(GOLD, "dit_small_golden.npz"))
    cfg = json.loads(bytes(g["cfg_json"]).decode())
    model = DiT(**cfg)
    args = [torch.from_numpy(g[k]) for k in ("x", "t", "cond_images", "static_latent", "xyz")]
    with pytest.raises(_lib.GvfError):
        model(*args)


def test_attention_backend_seam():
    import gvfdiffusion_amd.model.attention as A
    assert A.BACKEND == "hip"
    A.set_backend("hip")
    for other in ("flash_attn", "xformers", "sdpa", "naive"):
        with pytest.raises(ValueError):
            A.set_backend(other)


def test_options_the_reference_cannot_run_are_refused_at_construction():
    """share_mod=True and pe_mode='rope' end in shape errors inside the reference's own forward (profiles/r04_reference_dit_variants.txt,
    scripts/reference_dit_variants.py): there is no behaviour to match, the module says so instead of failing in a kernel; a head_dim other
    than 32 or 64 (here 16) has no HIP attention path, while one head of 64 is accepted (round 6)."""
    g = np.load(os.path.join(GOLD, "dit_small_golden.npz"))
    cfg = json.loads(bytes(g["cfg_json"]).decode())
    for over, word in ((dict(share_mod=True), "model/dit.py:247"), (dict(pe_mode="rope"), "modules.py:36"), (dict(num_heads=4), "head_dim 32 and 64")):
        with pytest.raises(NotImplementedError, match=word):
            DiT(**dict(cfg, **over))
    assert DiT(**dict(cfg, num_heads=1)).head_dim == 64


def test_param_version_sees_every_kind_of_weight_change_on_the_next_call():
    """DiT._param_version keys the packed-weight caches, the modulation table and the captured graph (the kept slots missed a replaced
    submodule for up to 255 forwards).  No GPU needed: the key itself must change on the very next call after an in-place update, a Parameter
    assigned to an existing slot, a replaced / added submodule, a newly registered parameter and a None -> tensor parameter -- and must NOT change
    when nothing happened."""
    import copy
    g = np.load(os.path.join(GOLD, "dit_small_golden.npz"))
    cfg = json.loads(bytes(g["cfg_json"]).decode())
    model = DiT(**cfg)
    seen = [model._param_version()]

    def moved():
        v = model._param_version()
        changed = v != seen[-1]
        seen.append(v)
        return changed
    assert not moved() and not moved()
    with torch.no_grad():
        model.blocks[0].mlp.mlp[0].weight.mul_(1.5)
    assert moved() and not moved()
    lin = model.blocks[1].spatial_self_attn.to_out
    lin.weight = torch.nn.Parameter(lin.weight.detach() * 0.5)
    assert moved() and not moved()
    model.blocks[0].mlp = copy.deepcopy(model.blocks[0].mlp)                   # a replaced submodule (new storage addresses)
    assert moved() and not moved()
    lin.bias = None                                                             # a parameter that goes away ...
    assert moved() and not moved()
    lin.bias = torch.nn.Parameter(torch.zeros(lin.out_features))                # ... and comes back (None -> tensor)
    assert moved() and not moved()
    model.blocks[0].register_parameter("extra_gain", torch.nn.Parameter(torch.ones(3)))
    assert moved() and not moved()
    model.blocks[1].add_module("extra", torch.nn.Linear(4, 4))
    assert moved() and not moved()
    # a child deleted and replaced in one go by a module of the same shape (with id()s in the fingerprint a recycled address passed as
    # "unchanged" and the kept `_parameters` dicts went stale; the fingerprint now holds the children themselves and compares identities)
    for _ in range(8):
        old_w = model.blocks[1].extra.weight.data_ptr()
        del model.blocks[1]._modules["extra"]
        model.blocks[1].add_module("extra", torch.nn.Linear(4, 4))
        assert moved() and not moved()              # (even when the allocator hands the new weight the old one's address: `old_w`)
    # and the key is exactly what a full walk gives (slots registered as None are kept as (-1, 0) place holders)
    key = model._param_version()
    assert key[0][0] == "structure"               # (the structural epoch leads the key)
    assert tuple(e for e in key[1:] if e != (-1, 0)) == tuple((p._version, p.data_ptr()) for p in model.parameters())


def test_sampling_scope_holds_the_param_version_for_one_sample_call():
    """DPM_Solver.sample brackets itself with model_wrapper's sampling_scope: the DiT walks its parameters once per sample (hold_param_version)
    instead of once per evaluation -- host time that is exposed wherever the sampler waits for the device (the adaptive solver's step-size test).
    Inside the bracket the key is the held one; after it a weight change is seen on the next call, and an exception inside sample() releases it."""
    from gvfdiffusion_amd.model.dpmsolver import NoiseScheduleVP, model_wrapper, DPM_Solver
    from gvfdiffusion_amd.model.gaussian_diffusion import create_gaussian_diffusion
    g = np.load(os.path.join(GOLD, "dit_small_golden.npz"))
    model = DiT(**json.loads(bytes(g["cfg_json"]).decode()))
    ns = NoiseScheduleVP("discrete", betas=torch.from_numpy(create_gaussian_diffusion(noise_schedule="cosine", predict_type="v").betas))
    walks, inside = [], []
    real = type(model)._param_version

    def counting(self):
        walks.append("_pver_held" in self.__dict__)
        return real(self)
    type(model)._param_version = counting
    try:
        def fake_forward(x, t, **kw):                       # stands in for the HIP forward: asks for the key as every forward does
            inside.append(model._param_version())
            return torch.zeros_like(x)
        model.forward = fake_forward
        fn = model_wrapper(model, ns, model_type="v", model_kwargs={})
        solver = DPM_Solver(fn, ns, algorithm_type="dpmsolver++")
        solver.verbose = False
        x = torch.zeros((1, 2, 4, 3))
        solver.sample(x, steps=5, t_start=1.0, t_end=1 / 1000, order=2, method="multistep")
        assert len(inside) == 5 and all(v is inside[0] for v in inside)          # one held tuple for the whole call
        assert "_pver_held" not in model.__dict__
        before = real(model)
        with torch.no_grad():
            model.blocks[0].mlp.mlp[0].weight.mul_(2.0)
        assert real(model) != before                                              # released: the next call sees the update

        # holds nest: two samplers that share the instance each bracket their own call; the inner one's release must not drop the outer's hold
        model.hold_param_version(True)
        outer = model.__dict__["_pver_held"]
        model.hold_param_version(True)
        model.hold_param_version(False)
        assert model.__dict__.get("_pver_held") is outer
        model.hold_param_version(False)
        assert "_pver_held" not in model.__dict__
        model.hold_param_version(False)                                           # an unmatched release is harmless
        assert model.__dict__.get("_pver_depth", 0) == 0

        def boom(x, t, **kw):
            raise RuntimeError("forward failed")
        model.forward = boom
        with pytest.raises(RuntimeError, match="forward failed"):
            solver.sample(x, steps=5, t_start=1.0, t_end=1 / 1000, order=2, method="multistep")
        assert "_pver_held" not in model.__dict__
    finally:
        type(model)._param_version = real
