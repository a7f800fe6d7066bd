"""N > 1 path of bench.py on CPU: two gloo ranks shard the samples (rank r owns samples r::world) and
exchange finished uint8 frames with one all_gather_into_tensor per step, exactly the calls bench.py makes on
RCCL.  The renderer itself needs a GPU, so the per-rank frames here come from the CPU oracle at a tiny size."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from gvfdiffusion_amd import synthetic
    from rast_util import camera_block, oracle_render
    F, S = 2, 48
    attrs = synthetic.random_gaussians(400, sh_degree=1, seed=rank, scale_lo=0.01, scale_hi=0.05)   # sample `rank`
    frames = np.stack([oracle_render(oracle, attrs, camera_block(azi=30.0 * f), S, S, 1)["color"] for f in range(F)])
    u8 = (torch.from_numpy(frames).clamp(0, 1) * 255).to(torch.uint8)                               # as bench.py
    gathered = torch.empty((world * F, 3, S, S), dtype=torch.uint8)          # concatenated along dim 0, as bench.py
    dist.all_gather_into_tensor(gathered, u8)
    gathered = gathered.reshape(world, F, 3, S, S)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                                                       # max-over-ranks timing
    np.save(os.path.join(out_dir, f"g{rank}.npy"), gathered.numpy())
    np.save(os.path.join(out_dir, f"own{rank}.npy"), u8.numpy())
    assert float(t) == world
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_sharding_and_frame_gather(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert np.array_equal(g0, g1)                                   # every rank holds every sample's frames
    for r in range(world):
        assert np.array_equal(g0[r], np.load(tmp_path / f"own{r}.npy"))
    assert not np.array_equal(g0[0], g0[1])                         # different samples per rank (sharded, not replicated)


# ---- BASELINE configs[4]: the sampler is what shards -------------------------------------------------------------------
def _toy_denoiser(x, t_input, cond_images=None, static_latent=None, deformation_position_xyz=None):
    """A per-sample network with the DiT's keyword interface: no cross-sample term, like the real denoiser
    (inference_dpm_latent.py:168-273 has no cross-sample operation; CFG triples the batch WITHIN a sample)."""
    s = (t_input / 1000.0).reshape(-1, 1, 1, 1)
    return 0.7 * x * (1 - s) + 0.1 * torch.tanh(cond_images.mean(dim=(2, 3), keepdim=True)[..., :1] + x) \
        + 0.05 * static_latent.mean(dim=(1, 2)).reshape(-1, 1, 1, 1)


def _sample_batch(indices, method):
    """DPM-Solver sample of the given global sample indices as ONE batch (what a rank does with its shard)."""
    from gvfdiffusion_amd.model.dpmsolver import NoiseScheduleVP, model_wrapper, DPM_Solver
    from gvfdiffusion_amd.model.gaussian_diffusion import create_gaussian_diffusion
    ns = NoiseScheduleVP("discrete", betas=torch.from_numpy(create_gaussian_diffusion(noise_schedule="cosine", predict_type="v").betas))
    xs, conds = [], {"cond_images": [], "static_latent": [], "deformation_position_xyz": []}
    for i in indices:                                   # per-sample seeds: the shard a rank draws does not depend on world size
        g = torch.Generator().manual_seed(100 + i)
        xs.append(torch.randn((1, 3, 8, 4), generator=g))
        conds["cond_images"].append(torch.randn((1, 3, 5, 6), generator=g))
        conds["static_latent"].append(torch.randn((1, 7, 4), generator=g))
        conds["deformation_position_xyz"].append(torch.rand((1, 8, 3), generator=g))
    cond = {k: torch.cat(v) for k, v in conds.items()}
    uncond = dict(cond); uncond["cond_images"] = torch.zeros_like(cond["cond_images"])
    mf = model_wrapper(_toy_denoiser, ns, model_type="v", model_kwargs={}, guidance_type="classifier-free", guidance_scale=2.0,
                       guidance_scale2=1.5, condition=cond, unconditional_condition=uncond)
    return DPM_Solver(mf, ns, algorithm_type="dpmsolver++").sample(torch.cat(xs), steps=8, t_start=1.0, t_end=1 / 1000, order=2,
                                                                  skip_type="time_uniform", method=method)


def _sampler_worker(rank, world, port, out_dir, total):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = list(range(rank, total, world))              # rank r owns samples r::world, as bench.py / DESIGN section 4
    x0 = _sample_batch(mine, "multistep")
    gathered = torch.empty((world * len(mine),) + tuple(x0.shape[1:]))
    dist.all_gather_into_tensor(gathered, x0.contiguous())       # the path's one collective (frames in bench.py; latents here)
    if rank == 0:
        np.save(os.path.join(out_dir, "sharded.npy"), gathered.reshape(world, len(mine), *x0.shape[1:]).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_sampling_equals_single_process(tmp_path):
    """Rank-sharded DPM_Solver.sample (with three-way classifier-free guidance) == the same samples drawn in one process:
    sharding over the batch changes nothing but where a sample is computed."""
    world, total, port = 2, 4, _free_port()
    mp.spawn(_sampler_worker, args=(world, port, str(tmp_path), total), nprocs=world, join=True)
    sharded = np.load(tmp_path / "sharded.npy")                    # [rank][local index]
    sys.path.insert(0, ROOT)
    single = _sample_batch(list(range(total)), "multistep").numpy()
    for r in range(world):
        for j, i in enumerate(range(r, total, world)):
            assert np.allclose(sharded[r, j], single[i], rtol=0, atol=1e-6), (r, j, i)
    assert not np.allclose(single[0], single[1])
