"""CPU restatement (numpy, float32) of farthest point sampling -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference takes it from `torch_cluster.fps` (third-party wheel, unpinned, absent from /root/reference; call sites
utils/inference_utils.py:195, model/autoencoder.py, encode_latent.py), whose published algorithm this restates: per batch
element start from a given point, keep every point's squared distance to its nearest selected point, repeatedly take
the arg-max.  **Parity unpinned** (no reference test pins the wheel; with its default `random_start=True` the reference's
own results are not reproducible either); what is pinned is the property every FPS must have -- checked in
tests/test_points.py: the selected set is the greedy k-centre sequence (each pick maximises the distance to the picks
before it), independent of how the work is split."""
import numpy as np


def fps_indices(pos: np.ndarray, ptr, k, start) -> np.ndarray:
    """pos (N,3) float32; batch b = rows [ptr[b], ptr[b+1]); k[b] picks starting at row ptr[b] + start[b].
    Squared distances as ((dx*dx + dy*dy) + dz*dz) in binary32; ties -> lowest index.  Returns int64 row numbers."""
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    out = []
    for b in range(len(k)):
        P = pos[ptr[b]:ptr[b + 1]]
        n = P.shape[0]
        dist = np.full((n,), np.inf, dtype=np.float32)
        sel = int(start[b])
        for it in range(int(k[b])):
            out.append(ptr[b] + sel)
            if it + 1 == int(k[b]):
                break
            d = P - P[sel]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            dist = np.minimum(dist, d2.astype(np.float32))
            sel = int(np.argmax(dist))          # first maximum = lowest index
    return np.asarray(out, dtype=np.int64)
