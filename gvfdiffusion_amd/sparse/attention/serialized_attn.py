"""Serialized (space-filling-curve window) self attention over a SparseTensor
(model/sparse_attention/serialized_attn.py:15-193): voxels of each sample are ordered by a Z-order / Hilbert
code (csrc/vox2seq.hip), cut into windows of exactly `window_size` tokens that overlap by modular padding
(M >= sum L gathered rows), attention runs per window and only each window's "valid" span is written back."""
import math
from enum import Enum
from typing import *

import torch

from ..basic import SparseTensor
from .. import vox2seq
from .full_attn import packed_varlen_attention

__all__ = ["SerializeMode", "SerializeModes", "calc_serialization", "sparse_serialized_scaled_dot_product_self_attention"]


class SerializeMode(Enum):
    Z_ORDER = 0
    Z_ORDER_TRANSPOSED = 1
    HILBERT = 2
    HILBERT_TRANSPOSED = 3


SerializeModes = [SerializeMode.Z_ORDER, SerializeMode.Z_ORDER_TRANSPOSED, SerializeMode.HILBERT,
                  SerializeMode.HILBERT_TRANSPOSED]

_MODE = {SerializeMode.Z_ORDER: ("z_order", [0, 1, 2]), SerializeMode.Z_ORDER_TRANSPOSED: ("z_order", [1, 0, 2]),
         SerializeMode.HILBERT: ("hilbert", [0, 1, 2]), SerializeMode.HILBERT_TRANSPOSED: ("hilbert", [1, 0, 2])}


def calc_serialization(tensor, window_size: int, serialize_mode: SerializeMode = SerializeMode.Z_ORDER,
                       shift_sequence: int = 0, shift_window: Tuple[int, int, int] = (0, 0, 0)):
    """-> (fwd_indices [M], bwd_indices [sum L], seq_lens, seq_batch_indices): the reference's contract (:36-117)."""
    if serialize_mode not in _MODE:
        raise ValueError(f"Unknown serialize mode: {serialize_mode}")
    dev = tensor.coords.device
    sc = tensor.coords[:, 1:].clone()
    sc += torch.tensor(shift_window, dtype=torch.int32, device=dev).reshape(1, 3)
    mode, permute = _MODE[serialize_mode]
    code = vox2seq.encode(sc, mode=mode, permute=permute)
    fwd_all, bwd_all, seq_lens, seq_batch = [], [], [], []
    offset_out = 0
    for bi, s in enumerate(tensor.layout):
        n = s.stop - s.start
        nw = (n + window_size - 1) // window_size
        order = torch.sort(code[s.start:s.stop].long(), stable=True).indices
        if nw == 1:
            fwd_all.append(order + s.start)
            inv = torch.empty_like(order)
            inv[order] = torch.arange(n, device=dev)
            bwd_all.append(inv + offset_out)
            seq_lens.append(n)
            seq_batch.append(bi)
            offset_out += n
            continue
        valid = n / nw
        split = [math.floor(i * valid + shift_sequence) for i in range(nw + 1)]
        bwd = torch.zeros((n,), dtype=torch.int64, device=dev)
        off = 0
        for i in range(nw):
            mid = (i + 0.5) * valid + shift_sequence
            ps = math.floor(mid - 0.5 * window_size)
            rows = order[torch.arange(ps, ps + window_size, device=dev) % n]
            vs, ve = split[i], split[i + 1]
            off += vs - ps
            bwd.scatter_(0, rows[vs - ps:ve - ps], torch.arange(off, off + ve - vs, device=dev))
            off += ps + window_size - vs
            fwd_all.append(rows + s.start)
        seq_lens.extend([window_size] * nw)
        seq_batch.extend([bi] * nw)
        bwd_all.append(bwd + offset_out)
        offset_out += nw * window_size
    return torch.cat(fwd_all), torch.cat(bwd_all), seq_lens, seq_batch


def sparse_serialized_scaled_dot_product_self_attention(qkv: SparseTensor, window_size: int,
                                                        serialize_mode: SerializeMode = SerializeMode.Z_ORDER,
                                                        shift_sequence: int = 0,
                                                        shift_window: Tuple[int, int, int] = (0, 0, 0)) -> SparseTensor:
    assert len(qkv.shape) == 4 and qkv.shape[1] == 3, f"Invalid shape for qkv, got {qkv.shape}, expected [N, *, 3, H, C]"
    name = f"serialization_{serialize_mode}_{window_size}_{shift_sequence}_{shift_window}"
    cache = qkv.get_spatial_cache(name)
    if cache is None:
        cache = calc_serialization(qkv, window_size, serialize_mode, shift_sequence, shift_window)
        qkv.register_spatial_cache(name, cache)
    fwd, bwd, seq_lens, _ = cache
    q, k, v = qkv.feats[fwd].unbind(dim=1)
    out = packed_varlen_attention(q, k, v, seq_lens, seq_lens)
    return qkv.replace(out[bwd])
