"""ctypes bindings of include/gvf_dit.h (csrc/gemm.hip, attn.hip, elem.hip) on torch device tensors."""
import ctypes

import torch

from .. import _lib

_vp, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

EPI_STORE_BF16, EPI_GELU_BF16, EPI_STORE_F32, EPI_RESID_F32 = 0, 1, 2, 3

_lib.register({
    "gvf_gemm_bf16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "gvf_gemm_stats_parts": (_i, [_i]),
    "gvf_gemm_bf16_resid_stats": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "gvf_gemm_ln_bf16": (_i, [_vp, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gvf_attn_fwd_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i] + [ctypes.POINTER(_i64)] * 4 + [_i, _vp, _vp, _f, _vp]),
    "gvf_attn_varlen_fwd_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i] + [ctypes.POINTER(_i64)] * 4 + [_vp, _vp, _f, _vp]),
    "gvf_attn_pack_kv_bf16": (_i, [_vp, _i, _i64, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "gvf_attn_tiled_fwd_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, ctypes.POINTER(_i64), ctypes.POINTER(_i64), _i64, _i64,
                                    _vp, _i, _i, _vp, _vp]),
    "gvf_layernorm_modulate_bf16": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "gvf_cast_pad_bf16": (_i, [_vp, _i, _vp, _i, _i64, _i, _i, _vp]),
})


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    return _lib.current_stream(t.device)


def pad64(k: int) -> int:
    return (k + 63) // 64 * 64


def cast_pad_bf16(src: torch.Tensor, ld_dst: int = None, act: int = 0, out: torch.Tensor = None) -> torch.Tensor:
    """fp32 (rows, cols) -> bf16 (rows, ld_dst) zero-padded; act 1 = SiLU."""
    _lib.require_cuda(src)
    assert src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1
    rows, cols = src.shape
    ld_dst = cols if ld_dst is None else ld_dst
    if out is None:
        out = torch.empty((rows, ld_dst), dtype=torch.bfloat16, device=src.device)
    _lib.check(_lib.lib().gvf_cast_pad_bf16(_p(src), src.stride(0), _p(out), ld_dst, rows, cols, act, _stream(src)),
               "gvf_cast_pad_bf16")
    return out


def gemm_bf16(a: torch.Tensor, w: torch.Tensor, bias, out: torch.Tensor, epilogue: int, gate: torch.Tensor = None,
              gate_ld: int = 0, rows_per_group: int = 0, n: int = None):
    """out (M, >=N) <- epilogue(a (M,K) @ w (N,K)^T + bias).  a, w bf16 with K % 64 == 0."""
    _lib.require_cuda(a, w, out)
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0] if n is None else n
    assert w.shape[1] == K and out.stride(-1) == 1
    _lib.check(_lib.lib().gvf_gemm_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K,
                                        epilogue, _p(gate), gate_ld, rows_per_group, _stream(a)), "gvf_gemm_bf16")
    return out


def gemm_stats_parts(n: int) -> int:
    return int(_lib.lib().gvf_gemm_stats_parts(int(n)))


def gemm_resid_stats(a, w, bias, x, stats, gate=None, gate_ld=0, rows_per_group=0):
    """x (M, N) fp32 += gate * (a @ w^T + bias), and stats (M, parts(N), 2) <- per-row partial (sum, sum of squares) of the
    UPDATED x (input of gemm_ln_bf16)."""
    _lib.require_cuda(a, w, x, stats)
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dtype == torch.float32 and stats.dtype == torch.float32
    M, K = a.shape
    N = w.shape[0]
    assert stats.is_contiguous() and stats.numel() >= M * gemm_stats_parts(N) * 2
    _lib.check(_lib.lib().gvf_gemm_bf16_resid_stats(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(x), x.stride(0), M, N, K,
                                                    _p(gate), gate_ld, rows_per_group, _p(stats), _stream(a)), "gvf_gemm_bf16_resid_stats")
    return x


def gemm_ln_bf16(x, stats, n_part, w, bias, out, epilogue, eps=1e-6, ln_w=None, ln_b=None, shift=None, scale=None, mod_ld=0,
                 rows_per_group=0):
    """out <- epilogue((LayerNorm(x) * s + t) @ w^T + bias) with LN statistics from gemm_resid_stats; see include/gvf_dit.h."""
    _lib.require_cuda(x, stats, w, out)
    assert x.dtype == torch.float32 and w.dtype == torch.bfloat16 and x.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    _lib.check(_lib.lib().gvf_gemm_ln_bf16(_p(x), x.stride(0), _p(stats), int(n_part), float(eps), _p(ln_w), _p(ln_b), _p(shift), _p(scale),
                                           int(mod_ld), int(rows_per_group), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K,
                                           int(epilogue), _stream(x)), "gvf_gemm_ln_bf16")
    return out


def _s4(st, head_dim=32):
    st = tuple(int(x) for x in st)
    if len(st) == 3:
        st = st + (head_dim,)    # packed heads: head h starts head_dim elements after head h-1
    return (_i64 * 4)(*st)


def attention_bf16(q, k, v, out, n_outer, n_inner, Lq, Lk, H, q_strides, k_strides, v_strides, o_strides, gamma_q=None,
                   gamma_k=None, scale=None, v_transposed=False, head_dim=32):
    """Strided flash attention (head_dim 32 or 64).  *_strides = (outer, inner, seq[, head = head_dim]) in
    elements; v_transposed: v stored [..][head][d][key] with v_strides[2] the d stride (see include/gvf_dit.h)."""
    _lib.require_cuda(q, k, v, out)
    assert q.dtype == k.dtype == v.dtype == out.dtype == torch.bfloat16
    scale = head_dim ** -0.5 if scale is None else scale
    _lib.check(_lib.lib().gvf_attn_fwd_bf16(_p(q), _p(k), _p(v), _p(out), n_outer, n_inner, Lq, Lk, H, head_dim,
                                            _s4(q_strides, head_dim), _s4(k_strides, head_dim), _s4(v_strides, head_dim),
                                            _s4(o_strides, head_dim), int(bool(v_transposed)), _p(gamma_q), _p(gamma_k),
                                            float(scale), _stream(q)), "gvf_attn_fwd_bf16")
    return out


def attention_varlen_bf16(q, k, v, out, cu_q, cu_k, max_Lq, max_Lk, H, q_strides, k_strides, v_strides, o_strides,
                          gamma_q=None, gamma_k=None, scale=None, head_dim=32):
    """Packed variable-length attention: cu_q / cu_k int32 device tensors [n_seqs + 1]."""
    _lib.require_cuda(q, k, v, out, cu_q, cu_k)
    assert q.dtype == k.dtype == v.dtype == out.dtype == torch.bfloat16
    assert cu_q.dtype == torch.int32 and cu_k.dtype == torch.int32 and cu_q.numel() == cu_k.numel()
    scale = head_dim ** -0.5 if scale is None else scale
    _lib.check(_lib.lib().gvf_attn_varlen_fwd_bf16(_p(q), _p(k), _p(v), _p(out), cu_q.numel() - 1, _p(cu_q), _p(cu_k),
                                                   int(max_Lq), int(max_Lk), H, head_dim, _s4(q_strides, head_dim),
                                                   _s4(k_strides, head_dim), _s4(v_strides, head_dim),
                                                   _s4(o_strides, head_dim), _p(gamma_q), _p(gamma_k), float(scale),
                                                   _stream(q)), "gvf_attn_varlen_fwd_bf16")
    return out


LOG2E = 1.4426950408889634


def attention_pack_kv(kv: torch.Tensor, n_sets: int, L: int, H: int, k_col0: int, v_col0: int, scale: float = None,
                      gamma_k: torch.Tensor = None, out=None):
    """kv rows (n_sets * L, ld) fp32 or bf16 -> (k_tiles, v_tiles) uint8 device buffers in the tiled cache image of
    csrc/attn_xt.hip (K pre-multiplied by scale * log2 e, optional RMSNorm gain)."""
    _lib.require_cuda(kv)
    assert kv.dim() == 2 and kv.stride(1) == 1 and kv.dtype in (torch.float32, torch.bfloat16)
    n_tiles = (L + 63) // 64
    nbytes = n_sets * H * n_tiles * 4096
    if out is None:
        out = (torch.empty(nbytes, dtype=torch.uint8, device=kv.device), torch.empty(nbytes, dtype=torch.uint8, device=kv.device))
    kt, vt = out
    assert kt.numel() >= nbytes and vt.numel() >= nbytes
    scale = 32 ** -0.5 if scale is None else scale
    _lib.check(_lib.lib().gvf_attn_pack_kv_bf16(_p(kv), int(kv.dtype == torch.float32), kv.stride(0), k_col0, v_col0, n_sets, L, H,
                                                float(scale * LOG2E), _p(gamma_k), _p(kt), _p(vt), _stream(kv)),
               "gvf_attn_pack_kv_bf16")
    return kt, vt


def attention_tiled_bf16(q, k_tiles, v_tiles, out, n_outer, n_inner, Lq, Lk, H, q_strides, o_strides, kv_stride_outer,
                         kv_stride_inner, gamma_q=None, force_exact=False, fallback_counter=None):
    """Cross attention against a tiled K/V cache (head_dim 32); see include/gvf_dit.h."""
    _lib.require_cuda(q, k_tiles, v_tiles, out)
    assert q.dtype == torch.bfloat16 and out.dtype in (torch.bfloat16, torch.float32)
    _lib.check(_lib.lib().gvf_attn_tiled_fwd_bf16(_p(q), _p(k_tiles), _p(v_tiles), _p(out), n_outer, n_inner, Lq, Lk, H,
                                                  _s4(q_strides), _s4(o_strides), int(kv_stride_outer), int(kv_stride_inner),
                                                  _p(gamma_q), int(out.dtype == torch.float32), int(bool(force_exact)),
                                                  _p(fallback_counter), _stream(q)),
               "gvf_attn_tiled_fwd_bf16")
    return out


def layernorm_modulate_bf16(x, out, eps=1e-6, ln_w=None, ln_b=None, shift=None, scale=None, mod_ld=0, rows_per_group=0):
    """x fp32 (rows, C) -> out bf16 (rows, C): LN then affine (ln_w, ln_b) and/or adaLN (shift, scale views)."""
    _lib.require_cuda(x, out)
    assert x.dtype == torch.float32 and out.dtype == torch.bfloat16 and x.is_contiguous() and out.is_contiguous()
    rows, C = x.shape
    _lib.check(_lib.lib().gvf_layernorm_modulate_bf16(_p(x), _p(out), rows, C, float(eps), _p(ln_w), _p(ln_b), _p(shift),
                                                      _p(scale), mod_ld, rows_per_group, _stream(x)),
               "gvf_layernorm_modulate_bf16")
    return out
