"""Farthest point sampling (csrc/fps.hip through include/gvf_points.h) and the Gaussian-tensor glue around it.
CPU part: the oracle (oracle/points_ref.py) has the defining property of the algorithm; GPU part: index-exact parity."""
import numpy as np
import pytest
import torch

from oracle import points_ref


def _greedy_property(pos, idx):
    """every pick maximises the distance to the picks before it (the greedy k-centre sequence)"""
    P = pos[idx].astype(np.float64)
    for j in range(1, len(idx)):
        d_sel = np.min(((pos.astype(np.float64) - P[:j, None]) ** 2).sum(-1), axis=0)
        assert d_sel[idx[j]] >= d_sel.max() * (1 - 1e-5)


def test_oracle_is_the_greedy_k_centre_sequence():
    g = np.random.default_rng(0)
    pos = g.standard_normal((500, 3)).astype(np.float32)
    idx = points_ref.fps_indices(pos, [0, 500], [40], [7])
    assert idx[0] == 7 and len(set(idx.tolist())) == 40
    _greedy_property(pos, idx)
    # a tiny known answer: points on a line, start in the middle -> far end first (lowest index on a tie), then the other end
    line = np.stack([np.arange(9, dtype=np.float32), np.zeros(9, np.float32), np.zeros(9, np.float32)], 1)
    assert points_ref.fps_indices(line, [0, 9], [4], [4]).tolist() == [4, 0, 8, 2]
    # ragged batches return row numbers of the stacked array
    both = np.concatenate([pos[:100], line])
    out = points_ref.fps_indices(both, [0, 100, 109], [5, 3], [0, 4])
    assert out[:5].max() < 100 and out[5:].tolist() == [104, 100, 108]


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,ks", [([5000], [64]), ([4096, 4097, 1], [4096, 10, 1]), ([20000, 300], [512, 300]), ([262144], [512])])
def test_fps_matches_oracle_index_for_index(cuda, sizes, ks):
    from gvfdiffusion_amd.utils.points import fps_counts
    g = torch.Generator().manual_seed(sum(sizes))
    pos = torch.rand((sum(sizes), 3), generator=g) - 0.5
    ptr = [0]
    for s in sizes:
        ptr.append(ptr[-1] + s)
    start = [int(torch.randint(0, s, (1,), generator=g)) for s in sizes]
    got = fps_counts(pos.to(cuda), ptr, ks, start).cpu().numpy()
    ref = points_ref.fps_indices(pos.numpy(), ptr, ks, start)
    assert np.array_equal(got, ref)


@pytest.mark.gpu
def test_torch_cluster_signature_and_sample_gs(cuda):
    from gvfdiffusion_amd.utils import fps, sample_gs, pad_static_gs
    g = torch.Generator().manual_seed(1)
    gs = [torch.rand((3000, 14), generator=g).to(cuda), torch.rand((1800, 14), generator=g).to(cuda)]
    batch = torch.cat([torch.zeros(3000, dtype=torch.long), torch.ones(1800, dtype=torch.long)]).to(cuda)
    stacked = torch.cat(gs)
    ratio = torch.tensor([256 / 3000, 256 / 1800], device=cuda)
    idx = fps(stacked[:, :3], batch, ratio=ratio, random_start=False)
    assert idx.shape == (512,) and idx.dtype == torch.int64
    ref = points_ref.fps_indices(stacked[:, :3].cpu().numpy(), [0, 3000, 4800], [256, 256], [0, 0])
    assert np.array_equal(idx.cpu().numpy(), ref)
    s = sample_gs(gs, 256, random_start=False)
    assert s.shape == (2, 256, 14) and torch.equal(s.reshape(512, 14), stacked[idx])
    # random starts: a valid greedy sequence from wherever it began
    s2 = fps(stacked[:, :3], batch, ratio=ratio)
    pos0 = stacked[:3000, :3].cpu().numpy()
    _greedy_property(pos0, s2[:256].cpu().numpy()[:32])
    padded, lens = pad_static_gs(gs)
    assert padded.shape == (2, 3000, 14) and lens == [3000, 1800]
    assert torch.equal(padded[1, 1800:, 10], torch.ones(1200, device=cuda)) and float(padded[1, 1800:, :10].abs().sum()) == 0.0


@pytest.mark.gpu
def test_many_examples_are_split_into_resident_calls(cuda):
    """20 examples (> 16 per call) of which two are large: the wrapper splits them into calls of <= 16 examples and <= 1024
    workgroups; every example still matches the oracle index for index."""
    from gvfdiffusion_amd.utils.points import fps_counts
    g = np.random.default_rng(3)
    sizes = [700, 5000] * 9 + [300_000, 150_000]
    ks = [16] * 18 + [8, 8]
    pos = g.random((sum(sizes), 3), dtype=np.float32)
    ptr = np.concatenate([[0], np.cumsum(sizes)]).tolist()
    got = fps_counts(torch.from_numpy(pos).cuda(), ptr, ks, [0] * 20).cpu().numpy()
    want = points_ref.fps_indices(pos, ptr, ks, [0] * 20)
    assert np.array_equal(got, want)
