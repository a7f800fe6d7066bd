"""The per-tile half of the rasteriser's sort stage on its own: every segment must come out in
upstream's stable (tile, depth) order = ascending (depth bits, Gaussian id), for every size class and for depth
distributions that defeat the distribution sort (csrc/rast.hip tile_sort_buckets) and send the segment through the network."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 2, 63, 64, 65, 128, 129, 200, 256, 257, 511, 512, 513, 700, 768, 769, 1000, 1024, 1025, 1279, 1280, 1281, 1500,
         1536, 1537, 2047, 2048, 2049, 3000, 4095, 4096, 4097, 5000, 16384, 16385, 20000]


def _depths(kind, n, rng):
    if kind == "uniform":                       # the bench scene: depths spread over the tile's range
        return rng.uniform(0.8, 1.6, n).astype(np.float32)
    if kind == "clusters":                      # a front and a back surface, each 1 % of the range wide
        return np.where(rng.random(n) < 0.5, rng.normal(1.0, 0.002, n), rng.normal(1.5, 0.002, n)).astype(np.float32)
    if kind == "one_depth":                     # a wall facing the camera: every key ties, order = id
        return np.full(n, 1.25, np.float32)
    if kind == "few_depths":                    # heavy ties inside a spread
        return rng.choice(np.linspace(0.9, 1.4, 7, dtype=np.float32), n)
    if kind == "outlier":                       # one far splat stretches the range, the rest share 1e-5 of it
        d = rng.uniform(1.0, 1.00001, n).astype(np.float32)
        if n:
            d[rng.integers(n)] = 900.0
        return d
    if kind == "tiny_range":                    # neighbouring floats only
        base = np.float32(1.3).view(np.uint32)
        return (base + rng.integers(0, 3, n).astype(np.uint32)).view(np.float32)
    if kind == "any_bits":                      # not depths at all: arbitrary high words (negative floats, NaNs, infinities)
        return rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    raise KeyError(kind)


@pytest.mark.parametrize("seed", [0, 1])
@pytest.mark.parametrize("kind", ["uniform", "clusters", "one_depth", "few_depths", "outlier", "tiny_range", "any_bits"])
def test_every_segment_is_sorted_by_depth_then_id(cuda, kind, seed):
    import zlib
    from gvfdiffusion_amd.rasterizer import tile_sort_u64
    rng = np.random.default_rng(zlib.crc32(kind.encode()) + seed)          # (reproducible: str hashes are salted per process)
    keys, ranges, expect = [], [], []
    at = 0
    for n in SIZES:
        d = _depths(kind, n, rng)
        ids = rng.permutation(300_000)[:n].astype(np.uint32)            # unique inside the segment, in arbitrary order
        k = (d.view(np.uint32).astype(np.uint64) << np.uint64(32)) | ids.astype(np.uint64)
        keys.append(k)
        ranges.append((at, at + n))
        expect.append((np.sort(k) & np.uint64(0xffffffff)).astype(np.uint32))
        at += n
        pad = int(rng.integers(0, 5))                                    # gaps between segments are legal (and untouched)
        keys.append(np.zeros(pad, np.uint64))
        expect.append(np.full(pad, 0xffffffff, np.uint32))
        at += pad
    keys = np.concatenate(keys)
    got = tile_sort_u64(torch.from_numpy(keys.view(np.int64)).to(cuda),
                        torch.tensor(ranges, dtype=torch.int32, device=cuda)).cpu().numpy().view(np.uint32)
    want = np.concatenate(expect)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (kind, bad[:8], [r for r in ranges if r[0] <= bad[0] < r[1]])


def test_many_segments_of_the_bench_shape(cuda):
    """60 000 segments with the size histogram of the bench scene: mostly 513-1280 keys."""
    from gvfdiffusion_amd.rasterizer import tile_sort_u64
    rng = np.random.default_rng(5)
    sizes = rng.choice([0, 30, 100, 200, 400, 800, 1100, 1200, 1700], size=3000, p=[0.4, 0.1, 0.04, 0.06, 0.09, 0.13, 0.1, 0.05, 0.03])
    starts = np.concatenate([[0], np.cumsum(sizes)])
    total = int(starts[-1])
    d = rng.uniform(0.8, 1.6, total).astype(np.float32)
    ids = rng.integers(0, 262144, total).astype(np.uint64)               # collisions across segments are fine; make them unique inside
    seg = np.repeat(np.arange(len(sizes)), sizes)
    ids = (np.arange(total) - starts[seg]).astype(np.uint64) * np.uint64(131) % np.uint64(262144) + np.uint64(0) * ids
    k = (d.view(np.uint32).astype(np.uint64) << np.uint64(32)) | ids
    order = np.lexsort((k, seg))
    want = (k[order] & np.uint64(0xffffffff)).astype(np.uint32)
    ranges = np.stack([starts[:-1], starts[1:]], 1).astype(np.int32)
    got = tile_sort_u64(torch.from_numpy(k.view(np.int64)).to(cuda), torch.from_numpy(ranges).to(cuda)).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want)
