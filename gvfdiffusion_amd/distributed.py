"""Batch-sharded sampling over the GPUs of one node: one process per GPU, samples r::world on rank r, ONE collective.

What the reference does (inference_dpm_latent.py:142-159, 168-273): every batch is processed sample by sample -- DPM-Solver sampling,
de-normalisation, motion-VAE decode, rendering -- with no operation across samples; under `accelerate launch` all ranks even walk the SAME
samples (dataset/dataset_latent_inference.py:39-47 is built with shard=0, num_shards=1; `accelerator.prepare` on a generator is a no-op)
and write rank-prefixed files (utils/inference_utils.py:297).  What this module adds is the sharding BASELINE.json asks for: rank r owns
samples r, r + world, ...; every rank holds the full models (0.46 GB of 16-bit weights); nothing is exchanged while a sample is computed
(spatial attention couples a sample's tokens, temporal attention its frames, three-way guidance triples the batch WITHIN a sample), and
the finished uint8 frames are collected with one all-gather -- RCCL over xGMI (torch.distributed backend "nccl") on an MI355X node, gloo
in the CPU tests.  46 MB per 24-frame 800x800 sample: ~1 ms of link time against ~200 ms of sampling, so the job scales with the number
of samples per rank and nothing else.

The functions take the process group's rank / world from torch.distributed when it is initialised and fall back to (0, 1), so the same
caller code runs on one GPU without a launcher."""
import os
from typing import Callable, List, Optional, Sequence

import torch


def rank_world(group=None):
    """(rank, world) of the default (or given) process group; (0, 1) when torch.distributed is not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def init_from_env(device: Optional[torch.device] = None, backend: Optional[str] = None):
    """Join the job a launcher described in RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run, accelerate launch):
    "nccl" (= RCCL on ROCm) for a GPU device, "gloo" otherwise.  No-op without WORLD_SIZE > 1 or when already initialised.
    Returns (rank, world)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and dist.is_available() and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (hosts without the legacy mode fail in hipIpcGetMemHandle); launchers normally export it
        use_gpu = device is not None and torch.device(device).type == "cuda"
        kw = {"device_id": torch.device(device)} if use_gpu else {}
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), **kw)
    return rank_world()


def shard_indices(total: int, rank: Optional[int] = None, world: Optional[int] = None) -> List[int]:
    """Global sample indices rank `rank` owns: rank, rank + world, ... < total (round-robin, so consecutive samples -- which tend to
    cost the same -- land on different ranks and every rank's count differs by at most one)."""
    if rank is None or world is None:
        rank, world = rank_world()
    if not (0 <= rank < world) or total < 0:
        raise ValueError(f"shard_indices: rank {rank} of {world}, total {total}")
    return list(range(rank, total, world))


def shard_size(total: int, world: int) -> int:
    """Samples per rank the collective is sized for: ceil(total / world) (ranks that own one sample fewer pad with zeros)."""
    return (total + world - 1) // world


def gather_frames(local: torch.Tensor, total: Optional[int] = None, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The path's one collective: local (n_local, ...) -- this rank's finished samples in the order of shard_indices, normally uint8
    frames (T, 3, H, W) each -- -> (total, ...) on EVERY rank in global sample order.  One all_gather_into_tensor of
    shard_size(total, world) samples per rank (a rank that owns fewer pads its block with zeros; the pad is dropped here).
    `total` defaults to world * n_local.  Without a process group: returns `local` (total must then equal n_local)."""
    import torch.distributed as dist
    rank, world = rank_world(group)
    n_local = local.shape[0]
    total = world * n_local if total is None else int(total)
    if n_local != len(shard_indices(total, rank, world)):
        raise ValueError(f"gather_frames: rank {rank} holds {n_local} samples, its shard of {total} over {world} ranks is {len(shard_indices(total, rank, world))}")
    if world == 1:
        return local
    per = shard_size(total, world)
    block = local.contiguous()
    if n_local < per:
        block = torch.cat([block, block.new_zeros((per - n_local,) + tuple(local.shape[1:]))])
    if block.is_cuda and dist.get_backend(group) == "gloo":
        # a gloo group over GPU tensors (two ranks sharing one GPU box in a test: RCCL refuses two ranks on one device): stage through the host
        flat_h = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype)
        dist.all_gather_into_tensor(flat_h, block.cpu(), group=group)
        flat = flat_h.to(local.device)
    else:
        flat = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(flat, block, group=group)
    # flat is rank-major: [rank][local index j] holds global sample j * world + rank
    by_rank = flat.view((world, per) + tuple(local.shape[1:]))
    glob = by_rank.transpose(0, 1).reshape((per * world,) + tuple(local.shape[1:]))[:total]
    if out is not None:
        out.copy_(glob)
        return out
    return glob.contiguous()


def run_sharded(job: Callable[[int, int], object], total: int, device=None, in_flight: int = 1, group=None) -> List[object]:
    """Run job(slot, global_index) for every sample this rank owns and return the results in shard order.  in_flight > 1 on a GPU keeps
    that many independent samples running on their own HIP streams / host threads (utils/in_flight.py: each slot needs its own mutable
    state -- one DiT instance per slot); the results do not depend on it."""
    rank, world = rank_world(group)
    mine = shard_indices(total, rank, world)
    if in_flight > 1 and device is not None and torch.device(device).type == "cuda" and len(mine) > 1:
        from .utils.in_flight import run_in_flight
        return run_in_flight([lambda slot, i=i: job(slot, i) for i in mine], device, in_flight)
    return [job(0, i) for i in mine]


def sample_decode_render_sharded(chain: Optional[Callable[[int, int], torch.Tensor]], total: int, device=None, in_flight: int = 1, group=None,
                                 gather: bool = True, batch_chain: Optional[Callable[[List[int]], Sequence[torch.Tensor]]] = None):
    """BASELINE configs[4] as one call: `chain(slot, i)` produces sample i's finished frames -- sample (DPM_Solver over the DiT) ->
    de-normalise -> VAE decode -> render -> uint8 (T, 3, H, W), the chain of inference_dpm_latent.py:225-272 -- on the rank that owns
    it; then the one frame all-gather.  `batch_chain(indices)` instead of `chain`: the rank's WHOLE share in one call (one batched
    DPM_Solver.sample over its samples -- the multistep solver and the DiT are batch-transparent, every launch of the forward then covers all
    of them --, frames returned in the order of `indices`); with in_flight > 1 on a GPU the share is cut into that many batches, each run as
    batch_chain(indices, slot) on its own stream.  Returns (frames, mine): frames (total, T, 3, H, W) in global order on every
    rank (gather=True) or this rank's (n_local, T, 3, H, W) block; mine = this rank's global sample indices."""
    rank, world = rank_world(group)
    if total < world:
        # decided from (total, world) alone, i.e. identically on EVERY rank and before any work: a check on the owner-less ranks only would
        # leave the others waiting in the all-gather until the collective times out
        raise ValueError(f"sample_decode_render_sharded: {total} samples over {world} ranks leaves ranks without a sample; run with fewer ranks")
    if (chain is None) == (batch_chain is None):
        raise ValueError("sample_decode_render_sharded: exactly one of chain / batch_chain")
    mine = shard_indices(total, rank, world)
    if batch_chain is not None:
        n_groups = max(1, min(int(in_flight), len(mine)))
        if n_groups > 1 and device is not None and torch.device(device).type == "cuda":
            # the rank's share as `in_flight` batches, each on its own HIP stream / host thread (utils/in_flight.py: slot k needs its own
            # mutable state, i.e. its own DiT instance): batch_chain(indices, slot).  Contiguous groups, so the results concatenate in order.
            per = (len(mine) + n_groups - 1) // n_groups
            groups = [mine[k * per:(k + 1) * per] for k in range(n_groups) if mine[k * per:(k + 1) * per]]
            from .utils.in_flight import run_in_flight
            parts = run_in_flight([lambda slot, g=g: list(batch_chain(list(g), slot)) for g in groups], device, len(groups))
            res = [r for part in parts for r in part]
        else:
            res = list(batch_chain(list(mine)))
        if len(res) != len(mine):
            raise ValueError(f"sample_decode_render_sharded: batch_chain returned {len(res)} samples for {len(mine)} indices")
    else:
        res = run_sharded(chain, total, device, in_flight, group)
    local = torch.stack([r if torch.is_tensor(r) else r[0] for r in res])
    if not gather:
        return local, mine
    return gather_frames(local, total, group), mine
