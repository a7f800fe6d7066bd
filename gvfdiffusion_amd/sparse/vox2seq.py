"""Drop-in for the reference's `vox2seq` extension
(model/sparse_voxel_diffusion/vox2seq/vox2seq/__init__.py:9-49 over src/ext.cpp:5-9): same
encode(coords, permute, mode) / decode(code, permute, mode) and the four _C-level functions, backed by
csrc/vox2seq.hip.  int32 in, int32 out, 3 x 10-bit coordinates."""
import ctypes
from typing import List

import torch

from .. import _lib

_vp, _i, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
_lib.register({name: (_i, [_vp, _vp, _vp, _vp, _i64, _vp]) for name in
               ("gvf_z_order_encode", "gvf_z_order_decode", "gvf_hilbert_encode", "gvf_hilbert_decode")})


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _encode(fn, x, y, z):
    _lib.require_cuda(x, y, z)
    x, y, z = (t.int().contiguous() for t in (x, y, z))
    code = torch.empty_like(x)
    _lib.check(getattr(_lib.lib(), fn)(_p(x), _p(y), _p(z), _p(code), x.numel(), _lib.current_stream(x.device)), fn)
    return code


def _decode(fn, code):
    _lib.require_cuda(code)
    code = code.int().contiguous()
    x, y, z = (torch.empty_like(code) for _ in range(3))
    _lib.check(getattr(_lib.lib(), fn)(_p(code), _p(x), _p(y), _p(z), code.numel(), _lib.current_stream(code.device)), fn)
    return x, y, z


def z_order_encode(x, y, z):
    return _encode("gvf_z_order_encode", x, y, z)


def hilbert_encode(x, y, z):
    return _encode("gvf_hilbert_encode", x, y, z)


def z_order_decode(code):
    return _decode("gvf_z_order_decode", code)


def hilbert_decode(code):
    return _decode("gvf_hilbert_decode", code)


@torch.no_grad()
def encode(coords: torch.Tensor, permute: List[int] = [0, 1, 2], mode: str = "z_order") -> torch.Tensor:
    assert coords.shape[-1] == 3 and coords.ndim == 2, "Input coordinates must be of shape [N, 3]"
    x, y, z = (coords[:, permute[k]].int() for k in range(3))
    if mode == "z_order":
        return z_order_encode(x, y, z)
    if mode == "hilbert":
        return hilbert_encode(x, y, z)
    raise ValueError(f"Unknown encoding mode: {mode}")


@torch.no_grad()
def decode(code: torch.Tensor, permute: List[int] = [0, 1, 2], mode: str = "z_order") -> torch.Tensor:
    assert code.ndim == 1, "Input code must be of shape [N]"
    if mode == "z_order":
        coords = z_order_decode(code)
    elif mode == "hilbert":
        coords = hilbert_decode(code)
    else:
        raise ValueError(f"Unknown decoding mode: {mode}")
    return torch.stack([coords[permute.index(0)], coords[permute.index(1)], coords[permute.index(2)]], dim=-1)
