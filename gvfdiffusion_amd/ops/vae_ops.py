"""ctypes bindings of include/gvf_vae.h (csrc/vae.hip) on torch device tensors."""
import ctypes

import torch

from .. import _lib
from .dit_ops import dt_code, LP_DTYPES

_vp, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

_lib.register({
    "gvf_geglu_bf16": (_i, [_vp, _i, _vp, _i, _i64, _i, _vp]),
    "gvf_vae_query_embed_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _f, _f, _vp]),
    "gvf_vae_embed_bf16_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _f, _vp]),
    "gvf_geglu": (_i, [_i, _vp, _i, _vp, _i, _i64, _i, _vp]),
    "gvf_vae_embed": (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _f, _vp]),
})


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def geglu_bf16(x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """x bf16 / fp16 (rows, 2F) -> (rows, F) = x[:, :F] * gelu_erf(x[:, F:]) in x's type  (model/autoencoder.py:90-93)."""
    _lib.require_cuda(x)
    assert x.dtype in LP_DTYPES and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % 2 == 0
    rows, F = x.shape[0], x.shape[1] // 2
    if out is None:
        out = torch.empty((rows, F), dtype=x.dtype, device=x.device)
    assert out.dtype == x.dtype
    _lib.check(_lib.lib().gvf_geglu(dt_code(x.dtype), _p(x), x.stride(0), _p(out), out.stride(0), rows, F, _lib.current_stream(x.device)),
               "gvf_geglu")
    return out


geglu = geglu_bf16


def vae_query_embed_bf16(queries: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, omega: torch.Tensor,
                         eps_embed: float = 1e-5, eps_prenorm: float = 1e-6, out: torch.Tensor = None, dtype=torch.bfloat16) -> torch.Tensor:
    """queries fp32 (P, qdim) -> bf16 (P, C): LN(LN(Linear(q)) + LN(PointEmbed(q[:, :3])))  (autoencoder.py:392-394,561,80)."""
    _lib.require_cuda(queries, weight, bias, omega)
    assert queries.dtype == weight.dtype == bias.dtype == omega.dtype == torch.float32
    assert queries.is_contiguous() and weight.is_contiguous() and bias.is_contiguous() and omega.is_contiguous()
    P, qdim = queries.shape
    C = weight.shape[0]
    assert weight.shape[1] == qdim and bias.numel() == C and omega.numel() * 6 == C
    if out is None:
        out = torch.empty((P, C), dtype=dtype, device=queries.device)
    _lib.check(_lib.lib().gvf_vae_embed(dt_code(out.dtype), _p(queries), qdim, _p(weight), _p(bias), _p(omega), _p(out), None, P, C, float(eps_embed),
                                        float(eps_prenorm), _lib.current_stream(queries.device)), "gvf_vae_embed")
    return out


def vae_embed_bf16_f32(rows: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, omega: torch.Tensor, want_embed: bool = True,
                       eps_embed: float = 1e-5, eps_prenorm: float = 1e-6, dtype=torch.bfloat16):
    """rows fp32 (P, qdim) -> (bf16 (P, C) PreNorm-normalised operand, fp32 (P, C) embedding or None): the encoder's
    input_embedding + position_encoding (model/autoencoder.py:520-524)."""
    _lib.require_cuda(rows, weight, bias, omega)
    assert rows.dtype == weight.dtype == bias.dtype == omega.dtype == torch.float32 and rows.is_contiguous() and weight.is_contiguous()
    P, qdim = rows.shape
    C = weight.shape[0]
    out = torch.empty((P, C), dtype=dtype, device=rows.device)
    emb = torch.empty((P, C), dtype=torch.float32, device=rows.device) if want_embed else None
    _lib.check(_lib.lib().gvf_vae_embed(dt_code(dtype), _p(rows), qdim, _p(weight), _p(bias), _p(omega), _p(out), _p(emb), P, C, float(eps_embed),
                                        float(eps_prenorm), _lib.current_stream(rows.device)), "gvf_vae_embed")
    return out, emb
