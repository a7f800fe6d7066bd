"""numpy restatement of Pillow's 8-bit separable resampler -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The algorithm lives in a third-party dependency of the reference (Pillow; the reference calls
`Image.fromarray(rgb).resize((t, t), resample=Image.Resampling.LANCZOS)` at utils/inference_utils.py:283 and pastes / crops
to 512x512 at :284-296).  Restated from Pillow's published source, src/libImaging/Resample.c: lanczos_filter / sinc_filter,
precompute_coeffs (support = 3 * max(scale, 1), window [int(c - s + .5), int(c + s + .5)), weights normalised to sum 1),
normalize_coeffs_8bpc (22-bit fixed point, round half away from zero), ImagingResampleHorizontal_8bpc / Vertical_8bpc
(int32 accumulator starting at 1 << 21, >> 22, clamp to 0..255; horizontal pass first, uint8 in between).
Pinned by tests/test_resize.py against Pillow itself (bit-exact on random images over up- and down-scaling ratios)."""
import numpy as np

PREC = 22


def _lanczos(x):
    x = np.asarray(x, dtype=np.float64)
    def sinc(v):
        out = np.ones_like(v)
        nz = v != 0.0
        pv = v[nz] * np.pi
        out[nz] = np.sin(pv) / pv
        return out
    return np.where((x >= -3.0) & (x < 3.0), sinc(x) * sinc(x / 3.0), 0.0)


def coeffs(in_size, out_size):
    """-> list of (xmin, int32 weights) per output sample."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 3.0 * fscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size)
        w = _lanczos((np.arange(xmin, xmax) - center + 0.5) / fscale)
        ww = 0.0
        for v in w:                                   # Pillow sums sequentially
            ww += v
        if ww != 0.0:
            w = w / ww
        k = np.where(w < 0, -0.5 + w * (1 << PREC), 0.5 + w * (1 << PREC)).astype(np.int64)   # C cast: truncation toward zero
        out.append((xmin, k.astype(np.int32)))
    return out


def _pass_rows(img, out_size):
    """Resample the LAST axis of a uint8 array."""
    tab = coeffs(img.shape[-1], out_size)
    res = np.empty(img.shape[:-1] + (out_size,), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx, (xmin, k) in enumerate(tab):
        acc = (1 << (PREC - 1)) + (src[..., xmin:xmin + len(k)] * k.astype(np.int64)).sum(axis=-1)
        res[..., xx] = np.clip(acc >> PREC, 0, 255).astype(np.uint8)
    return res


def resize_lanczos(img, out_h, out_w):
    """img (..., H, W) uint8 -> (..., out_h, out_w): horizontal pass then vertical pass (each skipped at equal size)."""
    if img.shape[-1] != out_w:
        img = _pass_rows(img, out_w)
    if img.shape[-2] != out_h:
        img = np.swapaxes(_pass_rows(np.swapaxes(img, -1, -2), out_h), -1, -2)
    return np.ascontiguousarray(img)


def resize_pad_crop(img, target, out_size=512, pad_value=255):
    """utils/inference_utils.py:283-296 for a square frame (..., S, S) uint8."""
    r = resize_lanczos(img, target, target)
    if target < out_size:
        canvas = np.full(img.shape[:-2] + (out_size, out_size), pad_value, dtype=np.uint8)
        p = max(0, (out_size - target) // 2)
        canvas[..., p:p + target, p:p + target] = r
        return canvas
    o = (target - out_size) // 2
    return np.ascontiguousarray(r[..., o:o + out_size, o:o + out_size])
