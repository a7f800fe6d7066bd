"""The N > 1 path (gvfdiffusion_amd/distributed.py, which bench.py --gpus N drives over RCCL) on CPU: two gloo ranks shard the samples
(rank r owns samples r::world), compute them with no exchange, and collect the finished uint8 frames with the path's ONE collective.
The HIP renderer needs a GPU, so the per-sample frames here come from the CPU oracle at a tiny size."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir, total):
    """One rank of the product path: gvfdiffusion_amd.distributed.sample_decode_render_sharded over a chain whose renderer is the CPU oracle
    (the HIP renderer needs a GPU); `total` samples that do not divide evenly over the ranks."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from gvfdiffusion_amd import distributed as D
    assert D.rank_world() == (0, 1)                                   # before the group exists: single-process defaults
    assert D.init_from_env() == (rank, world) and D.rank_world() == (rank, world)
    rendered = []

    def chain(slot, i):
        rendered.append(i)
        return _oracle_frames(i)

    frames, mine = D.sample_decode_render_sharded(chain, total)
    assert mine == D.shard_indices(total) == list(range(rank, total, world)) == rendered
    local, _ = D.sample_decode_render_sharded(chain, total, gather=False)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                          # max-over-ranks timing, as bench.py
    np.save(os.path.join(out_dir, f"g{rank}.npy"), frames.numpy())
    np.save(os.path.join(out_dir, f"own{rank}.npy"), local.numpy())
    assert float(t) == world
    with pytest.raises(ValueError):                                   # a block that is not this rank's shard is refused
        D.gather_frames(torch.zeros((len(mine) + 1, 2)), total)
    dist.barrier()
    dist.destroy_process_group()


def _oracle_frames(i, F=2, S=48):
    """uint8 frames (F, 3, S, S) of sample i: per-sample seed, so what a sample looks like does not depend on who renders it."""
    import oracle
    from gvfdiffusion_amd import synthetic
    from rast_util import oracle_render
    attrs = synthetic.random_gaussians(400, sh_degree=1, seed=i, scale_lo=0.01, scale_hi=0.05)
    frames = np.stack([oracle_render(oracle, attrs, synthetic.camera_block(azi=30.0 * f), S, S, 1)["color"] for f in range(F)])
    return (torch.from_numpy(frames).clamp(0, 1) * 255).to(torch.uint8)


def test_shard_indices_and_single_process_defaults():
    from gvfdiffusion_amd import distributed as D
    assert D.shard_indices(8, 1, 4) == [1, 5] and D.shard_indices(3, 2, 4) == [2] and D.shard_indices(3, 3, 4) == []
    assert sorted(sum((D.shard_indices(11, r, 4) for r in range(4)), [])) == list(range(11))
    assert D.shard_size(11, 4) == 3 and D.shard_size(8, 8) == 1
    with pytest.raises(ValueError):
        D.shard_indices(4, 4, 4)
    x = torch.arange(6).reshape(3, 2)
    assert D.gather_frames(x) is x                                     # no process group: the local block is the whole job
    frames, mine = D.sample_decode_render_sharded(lambda slot, i: torch.full((2,), i), 3)
    assert mine == [0, 1, 2] and frames.tolist() == [[0, 0], [1, 1], [2, 2]]
    # the rank's whole share in ONE call (a batched sampler): same result; a wrong count or both / neither callable is refused
    frames, mine = D.sample_decode_render_sharded(None, 3, batch_chain=lambda idx: [torch.full((2,), i) for i in idx])
    assert mine == [0, 1, 2] and frames.tolist() == [[0, 0], [1, 1], [2, 2]]
    with pytest.raises(ValueError):
        D.sample_decode_render_sharded(None, 3, batch_chain=lambda idx: [torch.zeros(2)])
    with pytest.raises(ValueError):
        D.sample_decode_render_sharded(None, 3)


def test_two_rank_sample_sharding_and_frame_gather(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    world, total, port = 2, 3, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), total), nprocs=world, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert g0.shape[0] == total and np.array_equal(g0, g1)          # every rank holds every sample's frames, in global order
    for i in range(total):
        assert np.array_equal(g0[i], _oracle_frames(i).numpy())     # == the same samples rendered in this one process
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"own{r}.npy"), g0[r::world])
    assert not np.array_equal(g0[0], g0[1])                         # different samples per rank (sharded, not replicated)


def _wide_worker(rank, world, port, out_dir, total):
    """sample_decode_render_sharded at the widths of the node (4 and 8 ranks): even and uneven shards, every rank checks the gathered job
    against its own shard, rank 0 saves it."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from gvfdiffusion_amd import distributed as D
    D.init_from_env()
    chain = lambda slot, i: _oracle_frames(i, F=1, S=32)            # noqa: E731
    frames, mine = D.sample_decode_render_sharded(chain, total)
    assert mine == list(range(rank, total, world)) and frames.shape[0] == total
    for i in mine:
        assert torch.equal(frames[i], _oracle_frames(i, F=1, S=32))
    # the same job with every rank's share produced by ONE batched call (bench.py's default sharded mode): identical gathered frames
    fb, mb = D.sample_decode_render_sharded(None, total, batch_chain=lambda idx: [_oracle_frames(i, F=1, S=32) for i in idx])
    assert mb == mine and torch.equal(fb, frames)
    digest = torch.tensor([float(frames.to(torch.float64).sum())], dtype=torch.float64)
    lo, hi = digest.clone(), digest.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert float(lo) == float(hi)                                     # every rank holds the same gathered job
    with pytest.raises(ValueError):                                   # fewer samples than ranks: refused on EVERY rank, before any work
        D.sample_decode_render_sharded(chain, world - 1)              # (a refusal on the owner-less ranks only would hang the others)
    if rank == 0:
        np.save(os.path.join(out_dir, "wide.npy"), frames.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(4, 8), (8, 11)])
def test_node_wide_sample_sharding(tmp_path, world, total):
    """The node's widths on CPU (gloo): 4 ranks x 2 samples and 8 ranks over 11 samples (three ranks own two, five own one + a zero pad)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    mp.spawn(_wide_worker, args=(world, _free_port(), str(tmp_path), total), nprocs=world, join=True)
    got = np.load(tmp_path / "wide.npy")
    assert got.shape[0] == total
    for i in range(total):
        assert np.array_equal(got[i], _oracle_frames(i, F=1, S=32).numpy()), i


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
    from gvfdiffusion_amd import distributed as D
    D.init_from_env()
    # every owned sample on its own (batch 1 with three-way guidance inside), as the product path runs them
    res = D.run_sharded(lambda slot, i: _sample_batch([i], "multistep")[0], total)
    gathered = D.gather_frames(torch.stack(res), total)          # the path's one collective (frames in the product; latents here)
    if rank == 0:
        np.save(os.path.join(out_dir, "sharded.npy"), gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_sampling_equals_single_process(tmp_path):
    """Rank-sharded DPM_Solver.sample (with three-way classifier-free guidance) == the same samples drawn in one process as one batch:
    sharding over the batch changes nothing but where a sample is computed (5 samples over 2 ranks: uneven shards)."""
    world, total, port = 2, 5, _free_port()
    mp.spawn(_sampler_worker, args=(world, port, str(tmp_path), total), nprocs=world, join=True)
    sharded = np.load(tmp_path / "sharded.npy")                    # global sample order
    sys.path.insert(0, ROOT)
    single = _sample_batch(list(range(total)), "multistep").numpy()
    assert sharded.shape == single.shape
    for i in range(total):
        assert np.allclose(sharded[i], single[i], rtol=0, atol=1e-6), i
    assert not np.allclose(single[0], single[1])


@pytest.mark.gpu
def test_bench_n2_line_carries_the_multi_rank_record(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), on the one-GPU box: the two ranks share
    the GPU and talk over gloo (GVF_BENCH_BACKEND=gloo, a test aid of bench.py; RCCL needs one GPU per rank), reduced raster size, the full
    DiT / VAE sharded-sampling leg.  A FUNCTIONAL record of the N > 1 code path -- the per-step frame gather inside the timed region, the
    batch-sharded sampling of BASELINE configs[4], the max-over-ranks reductions -- not a performance figure: the line must say who took part
    (`rccl_ranks`, `backend`), carry every rank's own ms per NFE, and the gather's bytes and time."""
    import json
    import subprocess
    env = dict(os.environ, GVF_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gaussians", "32768", "--res", "256"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                          # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["collective"].startswith("every counted sample")
    assert d["rccl_ranks"] == 2 and d["backend"] == "gloo"
    assert len(d["per_rank_ms_per_nfe"]) == 2 and all(v > 0 for v in d["per_rank_ms_per_nfe"])
    e = d["end_to_end"]
    assert e["samples"] == 8 and e["samples_per_rank"] == 4
    assert d["gather"]["bytes_per_rank"] == 4 * 24 * 3 * 256 * 256 and d["gather"]["bytes_total"] == 2 * d["gather"]["bytes_per_rank"]
    assert d["gather"]["us"] > 0 and len(d["gather"]["per_rank_us"]) == 2
    assert max(e["per_rank"]["ms_per_nfe"]) == pytest.approx(e["ms_per_nfe_slowest_rank"], rel=1e-3)
    print("bench --gpus 2 over gloo:", json.dumps({k: d[k] for k in ("value", "rccl_ranks", "backend", "per_rank_ms_per_nfe", "gather")}))
