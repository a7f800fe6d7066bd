"""Variable-length attention over SparseTensors with the reference's five calling conventions
(model/sparse_attention/full_attn.py:90-215 == sparse/attention/full_attn.py ==
trellis/modules/sparse/attention/full_attn.py): (qkv) | (q, kv) | (q, k, v) with q / kv sparse or dense.
The reference builds cu_seqlens on the host and calls flash_attn_varlen_* / xformers BlockDiagonalMask; here the
packed token lists go straight to the gfx950 kernel (gvf_attn_varlen_fwd_bf16, head_dim 32 or 64)."""
from typing import *

import torch

from ..basic import SparseTensor
from ...ops import dit_ops, precision

__all__ = ["sparse_scaled_dot_product_attention", "packed_varlen_attention"]


def _cu(lens: List[int], device) -> torch.Tensor:
    return torch.tensor([0] + list(torch.tensor(lens).cumsum(0).tolist()), dtype=torch.int32, device=device)


def packed_varlen_attention(q, k, v, q_lens: List[int], kv_lens: List[int], gamma_q=None, gamma_k=None):
    """q [Tq,H,C], k/v [Tk,H,C] packed over sequences -> [Tq,H,C] (same dtype as q).  fp16 / bf16 inputs are contracted in their own type
    (the reference hands flash-attn whatever autocast produced: fp16 under accelerate's mixed_precision='fp16'); fp32 inputs take
    ops/precision.py's choice."""
    Tq, H, C = q.shape
    dt = q.dtype
    lp = precision.resolve(None, (q, k, v))
    q, k, v = (t.to(lp) for t in (q, k, v))
    q, k, v = (t if (t.stride(2) == 1 and t.stride(1) == C) else t.contiguous() for t in (q, k, v))
    out = torch.empty((Tq, H, C), dtype=lp, device=q.device)
    dit_ops.attention_varlen(q, k, v, out, _cu(q_lens, q.device), _cu(kv_lens, q.device), max(q_lens), max(kv_lens), H,
                                  (0, 0, q.stride(0)), (0, 0, k.stride(0)), (0, 0, v.stride(0)), (0, 0, out.stride(0)),
                                  gamma_q, gamma_k, head_dim=C)
    return out.to(dt)


def sparse_scaled_dot_product_attention(*args, **kwargs):
    arg_names = {1: ["qkv"], 2: ["q", "kv"], 3: ["q", "k", "v"]}
    n = len(args) + len(kwargs)
    assert n in arg_names, f"Invalid number of arguments, got {n}, expected 1, 2, or 3"
    vals = list(args)
    for key in arg_names[n][len(args):]:
        assert key in kwargs, f"Missing argument {key}"
        vals.append(kwargs[key])

    def lens_of(t, L=None):
        return [t.layout[i].stop - t.layout[i].start for i in range(t.shape[0])] if isinstance(t, SparseTensor) else [L] * t.shape[0]

    if n == 1:
        qkv = vals[0]
        assert isinstance(qkv, SparseTensor), f"qkv must be a SparseTensor, got {type(qkv)}"
        assert len(qkv.shape) == 4 and qkv.shape[1] == 3, f"Invalid shape for qkv, got {qkv.shape}, expected [N, *, 3, H, C]"
        s = qkv
        q_lens = kv_lens = lens_of(qkv)
        q, k, v = qkv.feats.unbind(dim=1)                                  # [T, H, C] strided views
    elif n == 2:
        q, kv = vals
        assert q.shape[0] == kv.shape[0], f"Batch size mismatch, got {q.shape[0]} and {kv.shape[0]}"
        s = q if isinstance(q, SparseTensor) else None
        q_lens = lens_of(q, None if isinstance(q, SparseTensor) else q.shape[1])
        kv_lens = lens_of(kv, None if isinstance(kv, SparseTensor) else kv.shape[1])
        N = q.shape[0]
        q = q.feats if isinstance(q, SparseTensor) else q.reshape(-1, *q.shape[2:])
        kvf = kv.feats if isinstance(kv, SparseTensor) else kv.reshape(-1, *kv.shape[2:])
        k, v = kvf.unbind(dim=1)
    else:
        q, k, v = vals
        assert q.shape[0] == k.shape[0] == v.shape[0], "Batch size mismatch"
        s = q if isinstance(q, SparseTensor) else None
        q_lens = lens_of(q, None if isinstance(q, SparseTensor) else q.shape[1])
        kv_lens = lens_of(k, None if isinstance(k, SparseTensor) else k.shape[1])
        N = q.shape[0]
        q = q.feats if isinstance(q, SparseTensor) else q.reshape(-1, *q.shape[2:])
        k = k.feats if isinstance(k, SparseTensor) else k.reshape(-1, *k.shape[2:])
        v = v.feats if isinstance(v, SparseTensor) else v.reshape(-1, *v.shape[2:])
    out = packed_varlen_attention(q, k, v, q_lens, kv_lens)
    if s is not None:
        return s.replace(out)
    return out.reshape(N, q_lens[0], *out.shape[1:])
