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
