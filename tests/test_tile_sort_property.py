"""The exactness argument of the per-tile distribution sort (csrc/rast.hip tile_sort_buckets), checked on the CPU in the kernel's own
float32 arithmetic: the bucket of a key is a monotone function of the key's high word (the depth bits taken as an unsigned integer), so
keys of a lower bucket sort before keys of a higher one whatever the bit patterns are, and bucket order + exact ranks inside a bucket
give the sorted order."""
import numpy as np
from hypothesis import given, settings, strategies as st


def buckets(bits, nb):
    """min(NB - 1, int(float(bits - lo) * (NB / float(hi - lo)))) with every step rounded to float32 as the device does"""
    bits = np.asarray(bits, dtype=np.uint32)
    lo, hi = bits.min(), bits.max()
    scale = np.float32(nb) / np.float32(hi - lo) if hi > lo else np.float32(0.0)
    prod = (bits - lo).astype(np.float32) * scale                    # uint32 -> float32 rounds to nearest: monotone
    return np.minimum(nb - 1, prod.astype(np.int64)).astype(np.int64)


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(0, 2 ** 32 - 1), min_size=2, max_size=300), st.sampled_from([256, 768, 1280, 2048, 4096]))
def test_bucket_is_monotone_in_the_depth_bits(vals, nb):
    bits = np.array(sorted(vals), dtype=np.uint32)
    b = buckets(bits, nb)
    assert (np.diff(b) >= 0).all() and b.min() >= 0 and b.max() <= nb - 1


@settings(max_examples=50, deadline=None)
@given(st.integers(0, 2 ** 31), st.integers(1, 2000), st.sampled_from(["uniform", "clustered", "ties", "bits"]))
def test_bucket_order_plus_in_bucket_ranks_is_the_sorted_order(seed, n, kind):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        d = rng.uniform(0.8, 1.6, n).astype(np.float32).view(np.uint32)
    elif kind == "clustered":
        d = np.where(rng.random(n) < 0.5, rng.normal(1.0, 0.002, n), rng.normal(1.5, 0.002, n)).astype(np.float32).view(np.uint32)
    elif kind == "ties":
        d = rng.choice(np.linspace(0.9, 1.4, 5, dtype=np.float32), n).view(np.uint32)
    else:
        d = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    ids = rng.permutation(1 << 18)[:n].astype(np.uint64)
    keys = (d.astype(np.uint64) << np.uint64(32)) | ids
    nb = 256 * max(1, -(-n // 256))
    b = buckets(d, nb)
    # the kernel: position = bucket base (exclusive scan of the histogram) + number of smaller keys in the same bucket
    base = np.concatenate([[0], np.cumsum(np.bincount(b, minlength=nb))])
    pos = np.array([base[b[i]] + np.count_nonzero(keys[b == b[i]] < keys[i]) for i in range(n)])
    out = np.empty(n, np.uint64)
    out[pos] = keys
    assert np.array_equal(out, np.sort(keys))
