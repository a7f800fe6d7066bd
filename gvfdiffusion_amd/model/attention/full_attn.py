"""scaled_dot_product_attention with the reference's three calling conventions
(model/attention/full_attn.py:38-140): (qkv [N,L,3,H,C]) | (q [N,L,H,C], kv [N,L,2,H,C]) |
(q, k, v [N,L,H,C]); no mask, no dropout, scale 1/sqrt(C); returns [N,L,H,C] in the input dtype.
fp16 / bf16 tensors are contracted in their own type (fp16 MFMA / bf16 MFMA, fp32 softmax and accumulation) -- flash-attn's contract; fp32
tensors are rounded to the type ops/precision.py resolves (an active autocast region's dtype, else bf16)."""
import torch

from ...ops import dit_ops, precision

__all__ = ["scaled_dot_product_attention"]


def _strides(t):
    # [N, L, H, C] view: batch stride, (no inner), sequence stride; heads must be packed (stride C) and C contiguous
    assert t.stride(3) == 1 and t.stride(2) == t.shape[3], "heads must be contiguous (H*C packed)"
    return (t.stride(0), 0, t.stride(1))


def scaled_dot_product_attention(*args, **kwargs):
    arg_names = {1: ["qkv"], 2: ["q", "kv"], 3: ["q", "k", "v"]}
    n = len(args) + len(kwargs)
    assert n in arg_names, f"Invalid number of arguments, got {n}, expected 1, 2, or 3"
    vals = list(args)
    for key in arg_names[n][len(args):]:
        assert key in kwargs, f"Missing argument {key}"
        vals.append(kwargs[key])
    if n == 1:
        qkv = vals[0]
        assert qkv.dim() == 5 and qkv.shape[2] == 3, f"Invalid shape for qkv, got {qkv.shape}, expected [N, L, 3, H, C]"
        q, k, v = qkv.unbind(dim=2)
    elif n == 2:
        q, kv = vals
        assert q.shape[0] == kv.shape[0], f"Batch size mismatch, got {q.shape[0]} and {kv.shape[0]}"
        assert q.dim() == 4, f"Invalid shape for q, got {q.shape}, expected [N, L, H, C]"
        assert kv.dim() == 5, f"Invalid shape for kv, got {kv.shape}, expected [N, L, 2, H, C]"
        k, v = kv.unbind(dim=2)
    else:
        q, k, v = vals
        assert q.shape[0] == k.shape[0] == v.shape[0], "Batch size mismatch"
        assert q.dim() == 4 and k.dim() == 4 and v.dim() == 4, "expected [N, L, H, C] tensors"
    from . import BACKEND
    if BACKEND != "hip":
        raise ValueError(f"Unknown attention module: {BACKEND}")
    N, Lq, H, C = q.shape
    if C not in (32, 64):
        raise NotImplementedError(f"hip attention backend: head_dim {C} (32 and 64 are built)")
    dt = q.dtype
    lp = precision.resolve(tensors=(q, k, v))
    q, k, v = (t if t.dtype == lp else t.to(lp) for t in (q, k, v))
    q, k, v = (t if (t.stride(3) == 1 and t.stride(2) == C) else t.contiguous() for t in (q, k, v))
    out = torch.empty((N, Lq, H, C), dtype=lp, device=q.device)
    dit_ops.attention(q, k, v, out, N, 1, Lq, k.shape[1], H, _strides(q), _strides(k), _strides(v), _strides(out), head_dim=C)
    return out if dt == lp else out.to(dt)
