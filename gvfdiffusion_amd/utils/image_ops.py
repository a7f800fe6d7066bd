"""Frame post-process on the device: the resize -> pad / crop -> 512x512 step of render_and_save_images
(utils/inference_utils.py:276-297), bit-identical to the Pillow calls it replaces (`Image.resize(..., LANCZOS)`,
`Image.new` + `paste`, `crop`).  The host computes Pillow's coefficient tables (precompute_coeffs + normalize_coeffs_8bpc of
src/libImaging/Resample.c: a few thousand doubles per size pair, cached); the passes over the pixels run in csrc/resize.hip."""
import ctypes
import functools
import math
import torch

from .. import _lib

__all__ = ["resample_table", "resize_pad_crop_u8", "FILTERS"]

_i, _vp, _i64 = ctypes.c_int, ctypes.c_void_p, ctypes.c_int64


class _Table(ctypes.Structure):
    _fields_ = [("first", _vp), ("count", _vp), ("coef", _vp), ("ksize", ctypes.c_int32), ("n_out", ctypes.c_int32)]


_lib.register({"gvf_resample_place_u8": (_i, [_vp, _i64, _i, _i, ctypes.POINTER(_Table), ctypes.POINTER(_Table), _vp, _vp, _i, _i, _i, _i,
                                             _i, _vp])})

PRECISION_BITS = 22


def _sinc(x):
    if x == 0.0:
        return 1.0
    x *= math.pi
    return math.sin(x) / x


def _lanczos(x):
    return _sinc(x) * _sinc(x / 3) if -3.0 <= x < 3.0 else 0.0


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _box(x):
    return 1.0 if -0.5 < x <= 0.5 else 0.0


FILTERS = {"lanczos": (_lanczos, 3.0), "bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0), "box": (_box, 0.5)}


@functools.lru_cache(maxsize=64)
def _coeffs(in_size: int, out_size: int, filt: str):
    """Pillow's precompute_coeffs (box = the whole axis) + normalize_coeffs_8bpc -> (first[], count[], coef[ksize][out], ksize)."""
    fn, support = FILTERS[filt]
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = support * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    first, count = [], []
    coef = [[0] * out_size for _ in range(ksize)]
    ss = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [fn((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            coef[x][xx] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        first.append(xmin)
        count.append(xmax)
    return first, count, coef, ksize


@functools.lru_cache(maxsize=64)
def _device_table(in_size: int, out_size: int, filt: str, device_index: int):
    first, count, coef, ksize = _coeffs(in_size, out_size, filt)
    dev = torch.device("cuda", device_index)
    t = (torch.tensor(first, dtype=torch.int32, device=dev), torch.tensor(count, dtype=torch.int32, device=dev),
         torch.tensor(coef, dtype=torch.int32, device=dev).contiguous())
    return t, _Table(_lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), ksize, out_size)


def resample_table(in_size: int, out_size: int, filt: str = "lanczos"):
    """Host view of one pass table: (first, count, coef[ksize][out_size], ksize) as Python lists."""
    return _coeffs(int(in_size), int(out_size), filt)


def resize_pad_crop_u8(frames: torch.Tensor, target_size: int, out_size: int = 512, pad_value: int = 255,
                       filt: str = "lanczos", out: torch.Tensor = None) -> torch.Tensor:
    """frames (..., H, W) uint8 on the device -> (..., out_size, out_size): resize to target_size x target_size, then
    paste centred onto a `pad_value` canvas if smaller than out_size or centre-crop if larger, per axis exactly as
    utils/inference_utils.py:283-296 does with Pillow."""
    _lib.require_cuda(frames)
    if frames.dtype != torch.uint8 or frames.dim() < 2:
        raise ValueError("frames must be a uint8 tensor (..., H, W)")
    frames = frames.contiguous()
    H, W = frames.shape[-2:]
    target_size, out_size = int(target_size), int(out_size)
    if target_size < 1 or out_size < 1:
        raise ValueError("sizes must be positive")
    planes = frames.numel() // (H * W)
    lead = frames.shape[:-2]
    if out is None:
        out = torch.empty((*lead, out_size, out_size), dtype=torch.uint8, device=frames.device)
    elif out.shape != (*lead, out_size, out_size) or out.dtype != torch.uint8 or not out.is_contiguous():
        raise ValueError("out has the wrong shape / dtype")
    di = frames.device.index if frames.device.index is not None else torch.cuda.current_device()
    keep_h, tab_h = (None, None) if W == target_size else _device_table(W, target_size, filt, di)     # Image.resize copies at equal size
    keep_v, tab_v = (None, None) if H == target_size else _device_table(H, target_size, filt, di)
    # placement (:283-296): smaller than the canvas in EITHER axis -> paste at ((512-W)//2, (512-H)//2) on white, PIL clipping
    # whatever sticks out; otherwise centre crop.  The image is square here, so both axes take the same branch.
    off = max(0, (out_size - target_size) // 2) if target_size < out_size else -((target_size - out_size) // 2)
    step = 65535 // 1                                                     # planes per launch (gridDim.z)
    tmp = torch.empty((min(planes, step), H, (target_size + 3) // 4 * 4), dtype=torch.uint8, device=frames.device) if tab_h is not None else None
    src, dst = frames.view(planes, H, W), out.view(planes, out_size, out_size)
    for p0 in range(0, planes, step):
        n = min(step, planes - p0)
        _lib.check(_lib.lib().gvf_resample_place_u8(_lib.ptr(src[p0:]), n, H, W, ctypes.byref(tab_h) if tab_h is not None else None,
                                                    ctypes.byref(tab_v) if tab_v is not None else None, _lib.ptr(tmp), _lib.ptr(dst[p0:]),
                                                    out_size, out_size, off, off, int(pad_value), _lib.current_stream(frames.device)),
                   "gvf_resample_place_u8")
    return out
