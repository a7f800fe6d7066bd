"""ctypes bindings of include/gvf_dit.h (csrc/gemm.hip, attn.hip, attn_xt.hip, rowblock.hip, elem.hip) on torch device tensors.

The 16-bit operand type of a call (GVF_DT_BF16 / GVF_DT_F16 of the C ABI) is the torch dtype of its operand tensors: torch.bfloat16 or
torch.float16; buffers without a dtype of their own (the uint8 K / V^T tile images, packed weight streams) take it as an argument."""
import ctypes
import os

import torch

from .. import _lib

_vp, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

EPI_STORE_BF16, EPI_GELU_BF16, EPI_STORE_F32, EPI_RESID_F32, EPI_GEGLU_16 = 0, 1, 2, 3, 4
EPI_STORE_16, EPI_GELU_16 = EPI_STORE_BF16, EPI_GELU_BF16        # (the 16-bit output takes the call's operand type)
DT_BF16, DT_F16 = 0, 1
LP_DTYPES = (torch.bfloat16, torch.float16)


def dt_code(dtype) -> int:
    """torch.bfloat16 / torch.float16 -> GVF_DT_BF16 / GVF_DT_F16 (include/gvf_dit.h)."""
    if dtype == torch.bfloat16:
        return DT_BF16
    if dtype == torch.float16:
        return DT_F16
    raise _lib.GvfError(f"the matrix-pipe operand type must be torch.bfloat16 or torch.float16, got {dtype}")


def _same_lp(*tensors) -> int:
    dts = {t.dtype for t in tensors if t is not None and t.dtype in LP_DTYPES}
    if len(dts) != 1:
        raise _lib.GvfError(f"operands must share one 16-bit type (bf16 or fp16), got {[t.dtype for t in tensors if t is not None]}")
    return dt_code(dts.pop())



class RowblockLn(ctypes.Structure):
    """gvf_rowblock_ln of include/gvf_dit.h"""
    _fields_ = [("ln_w", _vp), ("ln_b", _vp), ("shift", _vp), ("scale", _vp)]


class RowblockArgs(ctypes.Structure):
    """gvf_rowblock_args of include/gvf_dit.h (same field order; native alignment)"""
    _fields_ = [("a", _vp), ("lda", ctypes.c_int32), ("K1", ctypes.c_int32), ("w", _vp), ("b1", _vp),
                ("x", _vp), ("M", ctypes.c_int32), ("C", ctypes.c_int32),
                ("x_in", _vp), ("x_in_period", ctypes.c_int32),
                ("in_x", _vp), ("in_wt", _vp), ("in_b", _vp), ("in_cin", ctypes.c_int32),
                ("gate1", _vp), ("ln1", RowblockLn),
                ("mod_ld", ctypes.c_int32), ("rows_per_group", ctypes.c_int32), ("eps", _f),
                ("b_fc1", _vp), ("b_fc2", _vp), ("hidden", ctypes.c_int32), ("gate_m", _vp), ("ln2", RowblockLn),
                ("b3", _vp), ("out3", _vp), ("N3", ctypes.c_int32), ("epi3", ctypes.c_int32),
                ("hb_out", _vp),
                ("k_tiles", _vp), ("v_tiles", _vp), ("kv_L", ctypes.c_int32), ("k_scale", _f), ("gamma_k", _vp),
                ("kv_group_rows", ctypes.c_int32), ("dtype", ctypes.c_int32),
                ("t_frames", ctypes.c_int32), ("t_stride", ctypes.c_int32),
                ("t_b_qkv", _vp), ("t_gamma_q", _vp), ("t_gamma_k", _vp), ("t_scale", _f),
                ("t_b_out", _vp), ("t_gate", _vp), ("t_ln", RowblockLn)]


_lib.register({
    "gvf_dit_timestep_embed_f32": (_i, [_vp, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "gvf_dit_modulation_f32": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "gvf_dit_input_layer_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "gvf_dit_final_layer_f32": (_i, [_vp, _i, _i, _f, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "gvf_dit_timestep_embed_bf16": (_i, [_vp, _i, _i, _f, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "gvf_rowblock_args_layout": (_i, [ctypes.POINTER(ctypes.c_int32), _i]),
    "gvf_rowblock_packed_bytes": (_i64, [_i, _i]),
    "gvf_rowblock_pack_weight": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "gvf_rowblock_pack_mlp": (_i, [_vp, _vp, _i, _vp, _vp]),
    "gvf_rowblock_fused": (_i, [ctypes.POINTER(RowblockArgs), _vp]),
    "gvf_rowblock_fused_bf16": (_i, [ctypes.POINTER(RowblockArgs), _vp]),
    "gvf_gemm_stats_parts": (_i, [_i]),
    "gvf_attn_key_order": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _vp]),
    "gvf_attn_pack_kv_ordered": (_i, [_i, _vp, _i, _i64, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "gvf_attn_key_order_groups": (_i, [_vp, _i64, _i64, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "gvf_attn_pack_kv_groups": (_i, [_i, _vp, _i, _i64, _i64, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "gvf_dpm_x0": (_i, [_vp, _vp, _f, _f, _vp, _i64, _vp]),
    "gvf_dpm_lincomb": (_i, [_vp, _vp, _vp, _f, _f, _f, _vp, _i64, _vp]),
    "gvf_dpm_err_scratch_doubles": (_i64, [_i, _i64]),
    "gvf_dpm_second_err": (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _vp, _vp, _i, _i64, _vp, _vp, _vp]),
    "gvf_split3_bf16": (_i, [_vp, _i64, _vp, _i64, _i, _i, _vp]),
    "gvf_attn_pack_kv64": (_i, [_i, _vp, _i, _i64, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "gvf_attn_tiled64_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, ctypes.POINTER(_i64), ctypes.POINTER(_i64), _i64, _i64, _i, _vp, _vp]),
    "gvf_gemm256_eligible": (_i, [_i, _i, _i, _i, _i, _i]),
    "gvf_gemm256": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "gvf_gemm8_eligible": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "gvf_gemm8": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gvf_attn_fold_pack": (_i, [_i, _vp, _i, _i, _i, _vp, _vp]),
    "gvf_attn_tiled64_fold_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, ctypes.POINTER(_i64), _i64, _i64, _i, _vp, _vp]),
    "gvf_attn_fold_reduce": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _vp]),
    "gvf_attn_tiled_fwd_pf": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, ctypes.POINTER(_i64), ctypes.POINTER(_i64), _i64, _i64,
                                   _vp, _i, _i, _vp, _vp, _i64, _vp]),
})
# every entry point that contracts 16-bit operands exists as NAME(dtype, ...) and as the round-1/2 wrapper NAME_bf16(...)
_GEMM_ARGS = [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]
_PAIRS = {
    ("gvf_gemm", "gvf_gemm_bf16"): _GEMM_ARGS,
    ("gvf_gemm_resid_stats", "gvf_gemm_bf16_resid_stats"): [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp],
    ("gvf_gemm_ln", "gvf_gemm_ln_bf16"): [_vp, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    ("gvf_attn_fwd", "gvf_attn_fwd_bf16"): [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i] + [ctypes.POINTER(_i64)] * 4 + [_i, _vp, _vp, _f, _vp],
    ("gvf_attn_varlen_fwd", "gvf_attn_varlen_fwd_bf16"): [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i] + [ctypes.POINTER(_i64)] * 4 + [_vp, _vp, _f, _vp],
    ("gvf_attn_pack_kv", "gvf_attn_pack_kv_bf16"): [_vp, _i, _i64, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp],
    ("gvf_attn_tiled_fwd", "gvf_attn_tiled_fwd_bf16"): [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, ctypes.POINTER(_i64), ctypes.POINTER(_i64), _i64, _i64,
                                                        _vp, _i, _i, _vp, _vp],
    ("gvf_layernorm_modulate", "gvf_layernorm_modulate_bf16"): [_vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _vp],
    ("gvf_cast_pad", "gvf_cast_pad_bf16"): [_vp, _i, _vp, _i, _i64, _i, _i, _vp],
}
_lib.register({new: (_i, [_i] + args) for (new, _old), args in _PAIRS.items()})
_lib.register({old: (_i, args) for (_new, old), args in _PAIRS.items()})


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    return _lib.current_stream(t.device)


def pad64(k: int) -> int:
    return (k + 63) // 64 * 64


def cast_pad(src: torch.Tensor, ld_dst: int = None, act: int = 0, out: torch.Tensor = None, dtype=torch.bfloat16) -> torch.Tensor:
    """fp32 (rows, cols) -> bf16 / fp16 (rows, ld_dst) zero-padded; act 1 = SiLU.  The type is out's, else `dtype`."""
    _lib.require_cuda(src)
    assert src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1
    rows, cols = src.shape
    ld_dst = cols if ld_dst is None else ld_dst
    if out is None:
        out = torch.empty((rows, ld_dst), dtype=dtype, device=src.device)
    _lib.check(_lib.lib().gvf_cast_pad(dt_code(out.dtype), _p(src), src.stride(0), _p(out), ld_dst, rows, cols, act, _stream(src)),
               "gvf_cast_pad")
    return out


def split3_bf16(src: torch.Tensor, weights: bool = False, out: torch.Tensor = None) -> torch.Tensor:
    """fp32 (rows, K) -> bf16 (rows, 3 * pad64(K)): the two-term expansion hi = bf16(x), lo = bf16(x - hi) laid out [hi | lo | hi] (activations) or
    [hi | hi | lo] (weights=True), so that gemm(split3(a), split3(w, True)) = a w^T to ~2^-16 on the bf16 matrix pipe; see include/gvf_dit.h."""
    _lib.require_cuda(src)
    assert src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1
    rows, K = src.shape
    if out is None:
        out = torch.empty((rows, 3 * pad64(K)), dtype=torch.bfloat16, device=src.device)
    assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.shape == (rows, 3 * pad64(K))
    _lib.check(_lib.lib().gvf_split3_bf16(_p(src), src.stride(0), _p(out), rows, K, int(bool(weights)), _stream(src)), "gvf_split3_bf16")
    return out


def cast_pad_bf16(src, ld_dst=None, act=0, out=None):
    return cast_pad(src, ld_dst, act, out, torch.bfloat16)


def gemm(a: torch.Tensor, w: torch.Tensor, bias, out: torch.Tensor, epilogue: int, gate: torch.Tensor = None,
              gate_ld: int = 0, rows_per_group: int = 0, n: int = None):
    """out (M, >=N) <- epilogue(a (M,K) @ w (N,K)^T + bias).  a, w bf16 or fp16 (the same) with K % 64 == 0; a 16-bit `out` has their type."""
    _lib.require_cuda(a, w, out)
    dt = _same_lp(a, w, out)
    assert a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0] if n is None else n
    assert w.shape[1] == K and out.stride(-1) == 1
    _lib.check(_lib.lib().gvf_gemm(dt, _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K,
                                   epilogue, _p(gate), gate_ld, rows_per_group, _stream(a)), "gvf_gemm")
    return out


def geglu_interleave(w: torch.Tensor, bias: torch.Tensor = None):
    """Rows of a GEGLU projection [value rows (F) | gate rows (F)] -> 64-row groups of 32 value rows then their 32 gate rows: the order
    EPI_GEGLU_16 expects (include/gvf_dit.h).  F % 32 == 0.  Returns (w', bias')."""
    F = w.shape[0] // 2
    assert w.shape[0] == 2 * F and F % 32 == 0
    idx = torch.arange(2 * F, device=w.device)
    blk, r = idx // 64, idx % 64
    src = torch.where(r < 32, blk * 32 + r, F + blk * 32 + (r - 32))
    return w[src].contiguous(), (None if bias is None else bias[src].contiguous())


def gemm_stats_parts(n: int) -> int:
    return int(_lib.lib().gvf_gemm_stats_parts(int(n)))


def gemm_resid_stats(a, w, bias, x, stats, gate=None, gate_ld=0, rows_per_group=0):
    """x (M, N) fp32 += gate * (a @ w^T + bias), and stats (M, parts(N), 2) <- per-row partial (sum, sum of squares) of the
    UPDATED x (input of gemm_ln_bf16)."""
    _lib.require_cuda(a, w, x, stats)
    dt = _same_lp(a, w)
    assert x.dtype == torch.float32 and stats.dtype == torch.float32
    M, K = a.shape
    N = w.shape[0]
    assert stats.is_contiguous() and stats.numel() >= M * gemm_stats_parts(N) * 2
    _lib.check(_lib.lib().gvf_gemm_resid_stats(dt, _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(x), x.stride(0), M, N, K,
                                               _p(gate), gate_ld, rows_per_group, _p(stats), _stream(a)), "gvf_gemm_resid_stats")
    return x


def gemm_ln_bf16(x, stats, n_part, w, bias, out, epilogue, eps=1e-6, ln_w=None, ln_b=None, shift=None, scale=None, mod_ld=0,
                 rows_per_group=0):
    """out <- epilogue((LayerNorm(x) * s + t) @ w^T + bias) with LN statistics from gemm_resid_stats; see include/gvf_dit.h."""
    _lib.require_cuda(x, stats, w, out)
    dt = _same_lp(w, out)
    assert x.dtype == torch.float32 and x.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    _lib.check(_lib.lib().gvf_gemm_ln(dt, _p(x), x.stride(0), _p(stats), int(n_part), float(eps), _p(ln_w), _p(ln_b), _p(shift), _p(scale),
                                      int(mod_ld), int(rows_per_group), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K,
                                      int(epilogue), _stream(x)), "gvf_gemm_ln")
    return out


def timestep_embed_bf16(t, w0, b0, w2, b2, out, freq_dim=256, max_period=10000.0, t_emb=None):
    """out <- bf16(silu(TimestepEmbedder(t))) (rows padded with zeros to out.shape[1]): the sinusoid, both Linears and both SiLUs of
    model/dit.py:59-100,217-225 in one launch.  w0 / w2: bf16 nn.Linear weights (K padded), b0 / b2 f32 or None."""
    _lib.require_cuda(t, w0, w2, out)
    assert t.dtype == torch.float32 and t.dim() == 1 and w0.dtype == w2.dtype == out.dtype == torch.bfloat16
    C = w2.shape[0]
    _lib.check(_lib.lib().gvf_dit_timestep_embed_bf16(_p(t), t.numel(), int(freq_dim), float(max_period), _p(w0), w0.stride(0), _p(b0), _p(w2),
                                                      w2.stride(0), _p(b2), C, _p(out), out.stride(0), _p(t_emb), _stream(t)),
               "gvf_dit_timestep_embed_bf16")
    return out


def timestep_embed_f32(t, w0, b0, w2, b2, freq_dim=256, max_period=10000.0, t_emb=None):
    """-> silu(TimestepEmbedder(t)) f32 (B, C), all in fp32 (w0 (C, freq_dim), w2 (C, C) fp32 nn.Linear weights)."""
    _lib.require_cuda(t, w0, w2)
    assert t.dtype == w0.dtype == w2.dtype == torch.float32 and w0.is_contiguous() and w2.is_contiguous()
    C = w2.shape[0]
    out = torch.empty((t.numel(), C), dtype=torch.float32, device=t.device)
    _lib.check(_lib.lib().gvf_dit_timestep_embed_f32(_p(t), t.numel(), int(freq_dim), float(max_period), _p(w0), _p(b0), _p(w2), _p(b2), C, _p(out),
                                                     _p(t_emb), _stream(t)), "gvf_dit_timestep_embed_f32")
    return out


def modulation_f32(s, w, bias, out=None):
    """out (B, N) = s (B, C) @ w (N, C)^T + bias, fp32 GEMV per sample (every adaLN projection of the step)."""
    _lib.require_cuda(s, w)
    assert s.dtype == w.dtype == torch.float32 and s.is_contiguous() and w.is_contiguous()
    B, C = s.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=s.device)
    _lib.check(_lib.lib().gvf_dit_modulation_f32(_p(s), B, C, _p(w), _p(bias), N, _p(out), _stream(s)), "gvf_dit_modulation_f32")
    return out


def input_layer_f32(x, w_t, bias, out, pos=None, pos_period=0, rows_per_group=0):
    """out (M, C) = pos (broadcast) + x (M, Cin) @ w_t (Cin, C) + bias, fp32; w_t = the nn.Linear weight transposed."""
    w = w_t
    _lib.require_cuda(x, w, out)
    assert x.dtype == w.dtype == out.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous() and out.is_contiguous()
    assert tuple(w.shape) == (x.shape[1], out.shape[1])
    _lib.check(_lib.lib().gvf_dit_input_layer_f32(_p(x), x.shape[0], x.shape[1], _p(w), _p(bias), _p(pos), int(pos_period), int(rows_per_group),
                                                  out.shape[1], _p(out), _stream(x)), "gvf_dit_input_layer_f32")
    return out


def final_layer_f32(x, w, bias, out, shift=None, scale=None, mod_ld=0, rows_per_group=0, eps=1e-6):
    """out (M, Cout) = (LayerNorm(x) * (1 + scale) + shift) @ w (Cout, 512)^T + bias, fp32, straight from the stream."""
    _lib.require_cuda(x, w, out)
    assert x.dtype == w.dtype == out.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous() and out.is_contiguous()
    _lib.check(_lib.lib().gvf_dit_final_layer_f32(_p(x), x.shape[0], x.shape[1], float(eps), _p(shift), _p(scale), int(mod_ld), int(rows_per_group),
                                                  _p(w), _p(bias), w.shape[0], _p(out), _stream(x)), "gvf_dit_final_layer_f32")
    return out


ROWBLOCK_C, ROWBLOCK_ROWS, ROWBLOCK_KPAD, ROWBLOCK_MAX_HIDDEN, ROWBLOCK_MAX_N3 = 512, 48, 128, 2048, 1536


def rowblock_supported(C: int, rows_per_group: int, hidden: int) -> bool:
    """The row-block kernel covers model_channels 512, 48-row blocks that do not straddle samples (callers pad a sample's rows to a
    multiple of 48: rowblock_padded_rows), an MLP of <= 2048 hidden units."""
    return C == ROWBLOCK_C and rows_per_group % ROWBLOCK_ROWS == 0 and hidden % ROWBLOCK_C == 0 and 0 < hidden <= ROWBLOCK_MAX_HIDDEN


def rowblock_padded_rows(rows: int) -> int:
    """Rows of a sample rounded up to whole 48-row blocks."""
    return (rows + ROWBLOCK_ROWS - 1) // ROWBLOCK_ROWS * ROWBLOCK_ROWS


def rowblock_pack_stream(w1, mlp=None, w3=None, temporal=None):
    """One weight stream for gvf_rowblock_fused: w1 = nn.Linear weight bf16 / fp16 [512][K1 padded to 128] (see cast_pad), mlp =
    (mlp.0 weight [hidden][512], mlp.2 weight [512][hidden]) or None, w3 = [N3][512] or None, all of ONE 16-bit type (the packers move 16-bit
    words: the stream has the type of its sources, which the launch must be told: rowblock_fused(dtype=...)).  temporal = (to_qkv weight
    [1536][512], to_out weight [512][512]) of the temporal section (instead of mlp).  Returns a uint8 tensor."""
    L = _lib.lib()
    ref = w1 if w1 is not None else (w3 if w3 is not None else mlp[0])
    _lib.require_cuda(ref)
    if w1 is not None:
        assert w1.dtype in LP_DTYPES and w1.shape[0] == ROWBLOCK_C and w1.shape[1] % ROWBLOCK_KPAD == 0 and w1.is_contiguous()
    sizes = [0 if w1 is None else int(L.gvf_rowblock_packed_bytes(ROWBLOCK_C, w1.shape[1]))]
    assert mlp is None or temporal is None
    if temporal is not None:
        tq, to = temporal
        assert tq.dtype == to.dtype and tq.dtype in LP_DTYPES and tq.is_contiguous() and to.is_contiguous()
        assert tuple(tq.shape) == (3 * ROWBLOCK_C, ROWBLOCK_C) and tuple(to.shape) == (ROWBLOCK_C, ROWBLOCK_C)
    sizes.append((0 if temporal is None else 4 * ROWBLOCK_C * ROWBLOCK_C * 2) if mlp is None else 2 * mlp[0].shape[0] * ROWBLOCK_C * 2)
    sizes.append(0 if w3 is None else int(L.gvf_rowblock_packed_bytes(w3.shape[0], ROWBLOCK_C)))
    out = torch.empty(sum(sizes), dtype=torch.uint8, device=ref.device)
    st = _stream(ref)
    if w1 is not None:
        _lib.check(L.gvf_rowblock_pack_weight(_p(w1), w1.stride(0), ROWBLOCK_C, w1.shape[1], _p(out), st), "gvf_rowblock_pack_weight")
    if mlp is not None:
        f1, f2 = mlp
        assert f1.dtype == f2.dtype and f1.dtype in LP_DTYPES and f1.is_contiguous() and f2.is_contiguous()
        assert f1.shape == (f2.shape[1], ROWBLOCK_C) and f2.shape[0] == ROWBLOCK_C
        _lib.check(L.gvf_rowblock_pack_mlp(_p(f1), _p(f2), f1.shape[0], _p(out[sizes[0]:]), st), "gvf_rowblock_pack_mlp")
    if temporal is not None:
        # the temporal section runs its passes v | q | k (csrc/rowblock.hip: v^T waits packed while q and k are projected), so the stream
        # carries to_qkv's rows in that order; the biases stay [b_q | b_k | b_v] (the kernel addresses them by name)
        tq = torch.cat([tq[2 * ROWBLOCK_C:], tq[:2 * ROWBLOCK_C]]).contiguous()
        _lib.check(L.gvf_rowblock_pack_weight(_p(tq), tq.stride(0), 3 * ROWBLOCK_C, ROWBLOCK_C, _p(out[sizes[0]:]), st), "gvf_rowblock_pack_weight")
        _lib.check(L.gvf_rowblock_pack_weight(_p(to), to.stride(0), ROWBLOCK_C, ROWBLOCK_C, _p(out[sizes[0] + 3 * ROWBLOCK_C * ROWBLOCK_C * 2:]), st),
                   "gvf_rowblock_pack_weight")
    if w3 is not None:
        assert w3.dtype in LP_DTYPES and w3.shape[1] == ROWBLOCK_C and w3.is_contiguous() and w3.shape[0] % ROWBLOCK_C == 0
        _lib.check(L.gvf_rowblock_pack_weight(_p(w3), w3.stride(0), w3.shape[0], ROWBLOCK_C, _p(out[sizes[0] + sizes[1]:]), st),
                   "gvf_rowblock_pack_weight")
    return out


def _pi(t):
    return None if t is None else int(t.data_ptr())       # ctypes.Structure pointer fields take ints


def _ln_struct(ln):
    ln = ln or {}
    return RowblockLn(_pi(ln.get("ln_w")), _pi(ln.get("ln_b")), _pi(ln.get("shift")), _pi(ln.get("scale")))


def rowblock_fused(a, stream_w, x, b1=None, gate1=None, ln1=None, mod_ld=0, rows_per_group=0, eps=1e-6,
                   mlp_bias=None, hidden=0, gate_m=None, ln2=None, b3=None, out3=None, hb_out=None, x_in=None, x_in_period=0, kv_tiles=None, kv_L=0, gamma_k=None, kv_scale=None, in_x=None, in_wt=None, in_b=None, kv_group_rows=0,
                   dtype=None, temporal=None):
    """x += gate1 * (a W1^T + b1); hb = LN1(x); [x += gate_m * MLP(hb); hb = LN2(x)]; out3 = hb W3^T + b3 or hb_out = hb -- ONE launch
    (csrc/rowblock.hip; include/gvf_dit.h).  ln1 / ln2: dict with ln_w, ln_b and / or shift, scale.  mlp_bias = (b_fc1, b_fc2).
    a = None: no closing projection (x already holds the sub-layer's result; the stream has no W1 segment).
    x_in (f32 [groups * x_in_period][C]): the residual is read from it, broadcast with period x_in_period inside a row group, and x is only written.
    temporal = dict(frames=T, stride=N, b_qkv, gamma_q, gamma_k, b_out, gate, ln[, scale]): the temporal self attention of the block
    between LN1 and the last projection (see include/gvf_dit.h; rows_per_group must be T * N)."""
    _lib.require_cuda(stream_w, x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    M, C = x.shape
    args = RowblockArgs()
    lp = [t for t in (a, out3, hb_out) if t is not None]
    dt = _same_lp(*lp) if lp else dt_code(dtype)          # the packed stream and the K / V^T tiles are of the same type (the caller's contract)
    assert dtype is None or dt_code(dtype) == dt
    args.dtype = dt
    if a is not None:
        _lib.require_cuda(a)
        assert a.stride(1) == 1
        args.a, args.lda, args.K1 = _pi(a), a.stride(0), a.shape[1]
    args.w, args.b1 = _pi(stream_w), _pi(b1)
    args.x, args.M, args.C = _pi(x), M, C
    if x_in is not None:
        assert x_in.dtype == torch.float32 and x_in.is_contiguous() and x_in.shape[-1] == C
        args.x_in, args.x_in_period = _pi(x_in), int(x_in_period)
    if in_x is not None:            # input_layer in fp32 inside the launch (a must be None)
        assert a is None and in_x.dtype == in_wt.dtype == torch.float32 and in_x.is_contiguous() and in_wt.is_contiguous()
        assert in_x.shape[0] == M and tuple(in_wt.shape) == (in_x.shape[1], C)
        args.in_x, args.in_wt, args.in_b, args.in_cin = _pi(in_x), _pi(in_wt), _pi(in_b), in_x.shape[1]
    args.gate1, args.ln1 = _pi(gate1), _ln_struct(ln1)
    args.mod_ld, args.rows_per_group, args.eps = int(mod_ld), int(rows_per_group), float(eps)
    if hidden:
        args.b_fc1, args.b_fc2 = _pi(mlp_bias[0]), _pi(mlp_bias[1])
        args.hidden, args.gate_m, args.ln2 = int(hidden), _pi(gate_m), _ln_struct(ln2)
    if out3 is not None:
        assert out3.is_contiguous() and out3.shape[0] == M
        args.b3, args.out3, args.N3, args.epi3 = _pi(b3), _pi(out3), out3.shape[1], EPI_STORE_BF16
    if kv_tiles is not None:        # to_qkv of the spatial self attention: out3 = q [M][C]; k, v -> tiled images (see attention_pack_kv)
        assert out3 is not None and out3.shape[1] == C and kv_L % 64 == 0
        if kv_group_rows:               # padded groups (see include/gvf_dit.h): rows_per_group rows each, the first kv_group_rows are tokens
            assert rows_per_group > 0 and M % rows_per_group == 0 and kv_group_rows % kv_L == 0 and kv_group_rows <= rows_per_group
            n_sets = (M // rows_per_group) * (kv_group_rows // kv_L)
            args.kv_group_rows = int(kv_group_rows)
        else:
            assert M % kv_L == 0
            n_sets = M // kv_L
        nbytes = n_sets * (C // 32) * (kv_L // 64) * 4096
        assert kv_tiles[0].numel() >= nbytes and kv_tiles[1].numel() >= nbytes
        args.N3 = 3 * C
        args.k_tiles, args.v_tiles, args.kv_L = _pi(kv_tiles[0]), _pi(kv_tiles[1]), int(kv_L)
        args.k_scale, args.gamma_k = float((32 ** -0.5 if kv_scale is None else kv_scale) * LOG2E), _pi(gamma_k)
    if hb_out is not None:
        assert hb_out.is_contiguous() and tuple(hb_out.shape) == (M, C)
        args.hb_out = _pi(hb_out)
    if temporal is not None:
        t = temporal
        args.t_frames, args.t_stride = int(t["frames"]), int(t["stride"])
        args.t_b_qkv, args.t_gamma_q, args.t_gamma_k = _pi(t.get("b_qkv")), _pi(t.get("gamma_q")), _pi(t.get("gamma_k"))
        args.t_scale = float(t.get("scale") or 32 ** -0.5)
        args.t_b_out, args.t_gate, args.t_ln = _pi(t.get("b_out")), _pi(t.get("gate")), _ln_struct(t.get("ln"))
    _lib.check(_lib.lib().gvf_rowblock_fused(ctypes.byref(args), _stream(x)), "gvf_rowblock_fused")
    return x


def _s4(st, head_dim=32):
    st = tuple(int(x) for x in st)
    if len(st) == 3:
        st = st + (head_dim,)    # packed heads: head h starts head_dim elements after head h-1
    return (_i64 * 4)(*st)


def attention(q, k, v, out, n_outer, n_inner, Lq, Lk, H, q_strides, k_strides, v_strides, o_strides, gamma_q=None,
                   gamma_k=None, scale=None, v_transposed=False, head_dim=32):
    """Strided flash attention (head_dim 32 or 64).  *_strides = (outer, inner, seq[, head = head_dim]) in
    elements; v_transposed: v stored [..][head][d][key] with v_strides[2] the d stride (see include/gvf_dit.h)."""
    _lib.require_cuda(q, k, v, out)
    dt = _same_lp(q, k, v, out)
    assert q.dtype == k.dtype == v.dtype == out.dtype
    scale = head_dim ** -0.5 if scale is None else scale
    _lib.check(_lib.lib().gvf_attn_fwd(dt, _p(q), _p(k), _p(v), _p(out), n_outer, n_inner, Lq, Lk, H, head_dim,
                                       _s4(q_strides, head_dim), _s4(k_strides, head_dim), _s4(v_strides, head_dim),
                                       _s4(o_strides, head_dim), int(bool(v_transposed)), _p(gamma_q), _p(gamma_k),
                                       float(scale), _stream(q)), "gvf_attn_fwd")
    return out


def attention_varlen(q, k, v, out, cu_q, cu_k, max_Lq, max_Lk, H, q_strides, k_strides, v_strides, o_strides,
                          gamma_q=None, gamma_k=None, scale=None, head_dim=32):
    """Packed variable-length attention: cu_q / cu_k int32 device tensors [n_seqs + 1]."""
    _lib.require_cuda(q, k, v, out, cu_q, cu_k)
    dt = _same_lp(q, k, v, out)
    assert q.dtype == k.dtype == v.dtype == out.dtype
    assert cu_q.dtype == torch.int32 and cu_k.dtype == torch.int32 and cu_q.numel() == cu_k.numel()
    scale = head_dim ** -0.5 if scale is None else scale
    _lib.check(_lib.lib().gvf_attn_varlen_fwd(dt, _p(q), _p(k), _p(v), _p(out), cu_q.numel() - 1, _p(cu_q), _p(cu_k),
                                              int(max_Lq), int(max_Lk), H, head_dim, _s4(q_strides, head_dim),
                                              _s4(k_strides, head_dim), _s4(v_strides, head_dim),
                                              _s4(o_strides, head_dim), _p(gamma_q), _p(gamma_k), float(scale),
                                              _stream(q)), "gvf_attn_varlen_fwd")
    return out


LOG2E = 1.4426950408889634


def key_order_by_norm(kv: torch.Tensor, n_sets: int, L: int, H: int, k_col0: int, n_first: int = 64) -> torch.Tensor:
    """int32 (n_sets, H, L): the n_first largest-norm keys of every (set, head) first, the rest behind them, both groups in context order --
    the order attention_pack_kv(key_order=...) stores a cross attention's cache in, so that the fp16 kernel's first-tile shift sees the
    high-norm keys (include/gvf_dit.h: gvf_attn_key_order, gvf_attn_pack_kv_ordered).  kv: fp32 rows; L <= 8192."""
    _lib.require_cuda(kv)
    assert kv.dtype == torch.float32 and kv.dim() == 2 and kv.stride(1) == 1
    order = torch.empty((n_sets, H, L), dtype=torch.int32, device=kv.device)
    _lib.check(_lib.lib().gvf_attn_key_order(_p(kv), kv.stride(0), k_col0, n_sets, L, H, int(n_first), _p(order), _stream(kv)), "gvf_attn_key_order")
    return order


def attention_pack_kv(kv: torch.Tensor, n_sets: int, L: int, H: int, k_col0: int, v_col0: int, scale: float = None,
                      gamma_k: torch.Tensor = None, out=None, dtype=None, key_order: torch.Tensor = None):
    """kv rows (n_sets * L, ld) fp32, bf16 or fp16 -> (k_tiles, v_tiles) uint8 device buffers in the tiled cache image of
    csrc/attn_xt.hip (K pre-multiplied by scale * log2 e, optional RMSNorm gain).  The tiles' 16-bit type is that of a 16-bit
    `kv`, else `dtype` (fp32 rows; default bf16).  key_order: optional int32 (n_sets, H, L) permutation of the keys per (set, head)."""
    _lib.require_cuda(kv)
    assert kv.dim() == 2 and kv.stride(1) == 1 and kv.dtype in (torch.float32,) + LP_DTYPES
    dt = dt_code(kv.dtype) if kv.dtype in LP_DTYPES else dt_code(dtype or torch.bfloat16)
    assert dtype is None or dt_code(dtype) == dt
    n_tiles = (L + 63) // 64
    nbytes = n_sets * H * n_tiles * 4096
    if out is None:
        out = (torch.empty(nbytes, dtype=torch.uint8, device=kv.device), torch.empty(nbytes, dtype=torch.uint8, device=kv.device))
    kt, vt = out
    assert kt.numel() >= nbytes and vt.numel() >= nbytes
    scale = 32 ** -0.5 if scale is None else scale
    if key_order is not None:
        assert key_order.dtype == torch.int32 and key_order.is_contiguous() and key_order.shape == (n_sets, H, L) and key_order.is_cuda
    _lib.check(_lib.lib().gvf_attn_pack_kv_ordered(dt, _p(kv), int(kv.dtype == torch.float32), kv.stride(0), k_col0, v_col0, n_sets, L, H,
                                                   float(scale * LOG2E), _p(gamma_k), _p(key_order), _p(kt), _p(vt), _stream(kv)),
               "gvf_attn_pack_kv_ordered")
    return kt, vt


def key_order_by_norm_groups(kv: torch.Tensor, n_groups: int, n_sets: int, L: int, H: int, k_col0: int, n_first: int = 64) -> torch.Tensor:
    """key_order_by_norm for n_groups row sets in one launch: kv fp32 (n_groups, n_sets * L, ld) -> int32 (n_groups, n_sets, H, L)."""
    _lib.require_cuda(kv)
    assert kv.dtype == torch.float32 and kv.dim() == 3 and kv.stride(2) == 1 and kv.shape[0] == n_groups and kv.shape[1] == n_sets * L
    order = torch.empty((n_groups, n_sets, H, L), dtype=torch.int32, device=kv.device)
    _lib.check(_lib.lib().gvf_attn_key_order_groups(_p(kv), kv.stride(1), kv.stride(0), n_groups, k_col0, n_sets, L, H, int(n_first), _p(order),
                                                    _stream(kv)), "gvf_attn_key_order_groups")
    return order


def attention_pack_kv_groups(kv: torch.Tensor, n_groups: int, n_sets: int, L: int, H: int, k_col0: int, v_col0: int, scale: float = None,
                             gamma_k: torch.Tensor = None, out=None, dtype=None, key_order: torch.Tensor = None):
    """attention_pack_kv for n_groups row sets of one shape in ONE launch: kv (n_groups, n_sets * L, ld) fp32 or 16-bit, gamma_k (n_groups, H * 32)
    fp32 or None, key_order int32 (n_groups, n_sets, H, L) or None -> (k_tiles, v_tiles) uint8 (n_groups, nbytes) -- row g is what
    attention_pack_kv(kv[g], ...) returns (bit for bit)."""
    _lib.require_cuda(kv)
    assert kv.dim() == 3 and kv.stride(2) == 1 and kv.dtype in (torch.float32,) + LP_DTYPES and kv.shape[0] == n_groups and kv.shape[1] == n_sets * L
    dt = dt_code(kv.dtype) if kv.dtype in LP_DTYPES else dt_code(dtype or torch.bfloat16)
    assert dtype is None or dt_code(dtype) == dt
    nbytes = n_sets * H * ((L + 63) // 64) * 4096
    if out is None:
        out = (torch.empty((n_groups, nbytes), dtype=torch.uint8, device=kv.device), torch.empty((n_groups, nbytes), dtype=torch.uint8, device=kv.device))
    kt, vt = out
    assert kt.shape == (n_groups, nbytes) and vt.shape == (n_groups, nbytes) and kt.is_contiguous() and vt.is_contiguous()
    scale = 32 ** -0.5 if scale is None else scale
    if gamma_k is not None:
        assert gamma_k.dtype == torch.float32 and gamma_k.is_contiguous() and gamma_k.shape == (n_groups, H * 32)
    if key_order is not None:
        assert key_order.dtype == torch.int32 and key_order.is_contiguous() and key_order.shape == (n_groups, n_sets, H, L) and key_order.is_cuda
    _lib.check(_lib.lib().gvf_attn_pack_kv_groups(dt, _p(kv), int(kv.dtype == torch.float32), kv.stride(1), kv.stride(0), n_groups, k_col0, v_col0,
                                                  n_sets, L, H, float(scale * LOG2E), _p(gamma_k), _p(key_order), _p(kt), _p(vt), _stream(kv)),
               "gvf_attn_pack_kv_groups")
    return kt, vt


def attention_pack_kv64(kv: torch.Tensor, n_sets: int, L: int, H: int, k_col0: int, v_col0: int, scale: float = None, out=None, dtype=None):
    """kv rows (n_sets * L, ld) fp32 or 16-bit -> (k_tiles, v_tiles) uint8 device buffers in the head_dim-64 image of csrc/attn_xt64.hip
    (K pre-multiplied by scale * log2 e; default scale 64 ** -0.5).  See include/gvf_dit.h."""
    _lib.require_cuda(kv)
    assert kv.dim() == 2 and kv.stride(1) == 1 and kv.dtype in (torch.float32,) + LP_DTYPES
    dt = dt_code(kv.dtype) if kv.dtype in LP_DTYPES else dt_code(dtype or torch.bfloat16)
    assert dtype is None or dt_code(dtype) == dt
    nbytes = n_sets * H * ((L + 63) // 64) * 8192
    if out is None:
        out = (torch.empty(nbytes, dtype=torch.uint8, device=kv.device), torch.empty(nbytes, dtype=torch.uint8, device=kv.device))
    kt, vt = out
    assert kt.numel() >= nbytes and vt.numel() >= nbytes
    scale = 64 ** -0.5 if scale is None else scale
    _lib.check(_lib.lib().gvf_attn_pack_kv64(dt, _p(kv), int(kv.dtype == torch.float32), kv.stride(0), k_col0, v_col0, n_sets, L, H,
                                             float(scale * LOG2E), _p(kt), _p(vt), _stream(kv)), "gvf_attn_pack_kv64")
    return kt, vt


def attention_tiled64(q, k_tiles, v_tiles, out, n_outer, n_inner, Lq, Lk, H, q_strides, o_strides, kv_stride_outer, kv_stride_inner,
                      force_exact=False, fallback_counter=None):
    """Cross attention against a tiled, LDS-resident K/V set (head_dim 64, Lk <= 512) of q's 16-bit type; see include/gvf_dit.h."""
    _lib.require_cuda(q, k_tiles, v_tiles, out)
    assert q.dtype in LP_DTYPES and out.dtype == q.dtype
    _lib.check(_lib.lib().gvf_attn_tiled64_fwd(dt_code(q.dtype), _p(q), _p(k_tiles), _p(v_tiles), _p(out), n_outer, n_inner, Lq, Lk, H,
                                               _s4(q_strides, 64), _s4(o_strides, 64), int(kv_stride_outer), int(kv_stride_inner),
                                               int(bool(force_exact)), _p(fallback_counter), _stream(q)), "gvf_attn_tiled64_fwd")
    return out


def gemm8(a: torch.Tensor, w: torch.Tensor, bias, out: torch.Tensor, epilogue: int = 0):
    """out = epi(a @ w.T + bias) on csrc/gemm8.hip's 256 x 256 x 64 tiles, eight waves (M, N multiples of 256, K of 64); epilogue EPI_STORE_16 or
    EPI_GEGLU_16 (out (M, N / 2)), EPI_RESID_F32 without a gate (out fp32 (M, N) += ..., 192-wide tiles: M, N multiples of 192), or EPI_STORE_F32
    (out fp32, ANY M).  gvf_gemm takes
    this kernel by itself for eligible shapes; this is the direct entry (tests, benchmarks)."""
    _lib.require_cuda(a, w, out)
    assert a.dtype in LP_DTYPES and w.dtype == a.dtype and out.dtype == (torch.float32 if epilogue in (EPI_RESID_F32, EPI_STORE_F32) else a.dtype)
    assert a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    _lib.check(_lib.lib().gvf_gemm8(dt_code(a.dtype), _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K, int(epilogue),
                                    _stream(a)), "gvf_gemm8")
    return out


def gemm8_eligible(M: int, N: int, K: int, lda: int, ldw: int, ldc: int, epilogue: int) -> int:
    """The tile gvf_gemm8 would run this call on (256 or 192), 0 = it refuses the shape (include/gvf_dit.h)."""
    return int(_lib.lib().gvf_gemm8_eligible(int(M), int(N), int(K), int(lda), int(ldw), int(ldc), int(epilogue)))


def gemm256(a: torch.Tensor, w: torch.Tensor, bias, out: torch.Tensor):
    """out = r16(a @ w.T + bias) on csrc/gemm256.hip's 256 x 256 x 64 tiles (opt-in kernel; M, N multiples of 256, K of 64; see include/gvf_dit.h)."""
    _lib.require_cuda(a, w, out)
    assert a.dtype in LP_DTYPES and w.dtype == a.dtype and out.dtype == a.dtype and a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    _lib.check(_lib.lib().gvf_gemm256(dt_code(a.dtype), _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K, _stream(a)),
               "gvf_gemm256")
    return out


def attention_fold_pack(w16: torch.Tensor, n_out: int, H: int):
    """(n_out <= 16, >= H*64) 16-bit row-major matrix -> the fragment image the fold epilogue of csrc/attn_xt64.hip reads (uint8, H * 4096 bytes)."""
    _lib.require_cuda(w16)
    assert w16.dim() == 2 and w16.stride(1) == 1 and w16.dtype in LP_DTYPES and w16.shape[0] >= n_out
    frags = torch.empty(H * 4096, dtype=torch.uint8, device=w16.device)
    _lib.check(_lib.lib().gvf_attn_fold_pack(dt_code(w16.dtype), _p(w16), w16.stride(0), n_out, H, _p(frags), _stream(w16)), "gvf_attn_fold_pack")
    return frags


def attention_tiled64_fold(q, k_tiles, v_tiles, fold_frags, part, n_outer, n_inner, Lq, Lk, H, q_strides, kv_stride_outer, kv_stride_inner,
                           force_exact=False, fallback_counter=None):
    """attention_tiled64 with the projection behind it folded into the epilogue: writes part (n_outer * n_inner, H, Lq, 16) fp32; see
    include/gvf_dit.h (gvf_attn_tiled64_fold_fwd)."""
    _lib.require_cuda(q, k_tiles, v_tiles, fold_frags, part)
    assert q.dtype in LP_DTYPES and part.dtype == torch.float32 and part.is_contiguous() and part.numel() >= n_outer * n_inner * H * Lq * 16
    _lib.check(_lib.lib().gvf_attn_tiled64_fold_fwd(dt_code(q.dtype), _p(q), _p(k_tiles), _p(v_tiles), _p(fold_frags), _p(part), n_outer, n_inner,
                                                    Lq, Lk, H, _s4(q_strides, 64), int(kv_stride_outer), int(kv_stride_inner),
                                                    int(bool(force_exact)), _p(fallback_counter), _stream(q)), "gvf_attn_tiled64_fold_fwd")
    return part


def attention_fold_reduce(part, bias, out, n_sets, H, Lq, n_out, out_set_stride, out_row_stride):
    """out[set][q][:n_out] = bias + sum over heads of part[set][head][q][:n_out] (fp32; `out` may be a view: strides in elements)."""
    _lib.require_cuda(part, out)
    assert part.dtype == torch.float32 and out.dtype == torch.float32 and (bias is None or bias.dtype == torch.float32)
    _lib.check(_lib.lib().gvf_attn_fold_reduce(_p(part), _p(bias), _p(out), n_sets, H, Lq, n_out, int(out_set_stride), int(out_row_stride),
                                               _stream(part)), "gvf_attn_fold_reduce")
    return out


def attention_tiled(q, k_tiles, v_tiles, out, n_outer, n_inner, Lq, Lk, H, q_strides, o_strides, kv_stride_outer,
                         kv_stride_inner, gamma_q=None, force_exact=False, fallback_counter=None, bounded=False, prefetch=None):
    """Cross attention against a tiled K/V cache (head_dim 32) of q's 16-bit type; see include/gvf_dit.h.  bounded: the caller vouches that
    every log2-domain score is <= 14 in magnitude (scores_bounded() below): fp16 then runs without the per-query shift.
    prefetch: optional tensor (the packed weight stream of the launch that follows): its bytes are touched once by this launch, so that the
    next launch finds them in the Infinity Cache (gvf_attn_tiled_fwd_pf); results do not depend on it."""
    _lib.require_cuda(q, k_tiles, v_tiles, out)
    assert q.dtype in LP_DTYPES and out.dtype in (q.dtype, torch.float32)
    pf_bytes = 0 if prefetch is None else prefetch.numel() * prefetch.element_size()
    _lib.check(_lib.lib().gvf_attn_tiled_fwd_pf(dt_code(q.dtype), _p(q), _p(k_tiles), _p(v_tiles), _p(out), n_outer, n_inner, Lq, Lk, H,
                                                _s4(q_strides), _s4(o_strides), int(kv_stride_outer), int(kv_stride_inner),
                                                _p(gamma_q), int(out.dtype == torch.float32), int(bool(force_exact)) | (2 if bounded else 0),
                                                _p(fallback_counter), _p(prefetch), pf_bytes, _stream(q)),
               "gvf_attn_tiled_fwd_pf")
    return out


ATTN_SCORE_BOUND = 15.5          # log2 units: exp2(15.5) < 65504, fp16's largest number (GVF_ATTN_SCORES_BOUNDED, include/gvf_dit.h)


def scores_bounded(gamma_q, gamma_k, head_dim=32, scale=None, strict=False) -> bool:
    """Can the fp16 tiled attention run without its per-query shift (bounded=True)?  With u, v the unit vectors of a query and a key of one
    head, the log2-domain score of MultiHeadRMSNorm'ed q and k is head_dim * sum_d u_d v_d gq_d gk_d * scale * log2 e.
    strict: the worst case over every u, v -- head_dim * max_d |gq_d gk_d| ... <= ATTN_SCORE_BOUND (all of a query's and a key's energy in the
    one channel with the largest gain product).  Default: the same expression with the root-mean-square of gq gk over a head's channels
    (what aligned u = v with evenly spread energy reach; gains of 1 give 8.2 against 15.5), maximum over the heads: a HINT, not a proof --
    the launch does not need one: an overflow (or a query whose scores all sit below -15) fails the kernel's range guard and its workgroup
    is recomputed with the running maximum (tests/test_dit_fp16_gpu.py::test_tiled_cache_attention_fp16_broken_bound_falls_back); a wrong
    hint costs time, never correctness.  Reads the two gain tensors on the host: call it once per weight version."""
    if gamma_q is None or gamma_k is None or os.environ.get("GVF_ATTN_BOUNDED", "1") == "0":      # (the switch: A/B measurements)
        return False
    scale = head_dim ** -0.5 if scale is None else scale
    g = (gamma_q.float() * gamma_k.float()).reshape(-1, head_dim)
    worst = float(g.abs().max()) if strict else float(g.square().mean(dim=1).sqrt().max())
    return head_dim * worst * scale * LOG2E * 1.01 <= ATTN_SCORE_BOUND


def layernorm_modulate(x, out, eps=1e-6, ln_w=None, ln_b=None, shift=None, scale=None, mod_ld=0, rows_per_group=0):
    """x fp32 (rows, C) -> out bf16 / fp16 (rows, C): LN then affine (ln_w, ln_b) and/or adaLN (shift, scale views)."""
    _lib.require_cuda(x, out)
    assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    rows, C = x.shape
    _lib.check(_lib.lib().gvf_layernorm_modulate(dt_code(out.dtype), _p(x), _p(out), rows, C, float(eps), _p(ln_w), _p(ln_b), _p(shift),
                                                 _p(scale), mod_ld, rows_per_group, _stream(x)),
               "gvf_layernorm_modulate")
    return out


# round-1/2 names (the functions take either 16-bit type)
gemm_bf16, attention_bf16, attention_varlen_bf16, attention_tiled_bf16 = gemm, attention, attention_varlen, attention_tiled
layernorm_modulate_bf16 = layernorm_modulate


# ---- DPM-Solver state updates (csrc/dpm.hip; include/gvf_dit.h) ------------------------------------------------------------------------
def dpm_fusable(*tensors) -> bool:
    """The solver's fused launches take contiguous fp32 device tensors of one shape."""
    t0 = tensors[0]
    return all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == t0.shape and t.device == t0.device
               for t in tensors)


def dpm_x0(x: torch.Tensor, noise: torch.Tensor, sigma: float, alpha: float) -> torch.Tensor:
    """x0 = (x - sigma * noise) / alpha in one launch (every operation rounded to fp32 on its own)."""
    out = torch.empty_like(x)
    _lib.check(_lib.lib().gvf_dpm_x0(_p(x), _p(noise), float(sigma), float(alpha), _p(out), x.numel(), _stream(x)), "gvf_dpm_x0")
    return out


def dpm_lincomb(x: torch.Tensor, m0: torch.Tensor, a: float, b: float, m1: torch.Tensor = None, c: float = 0.0) -> torch.Tensor:
    """((a x) + (b m0)) [+ (c m1)] in one launch."""
    out = torch.empty_like(x)
    _lib.check(_lib.lib().gvf_dpm_lincomb(_p(x), _p(m0), _p(m1), float(a), float(b), float(c), _p(out), x.numel(), _stream(x)), "gvf_dpm_lincomb")
    return out


def dpm_second_err(x, m, m1, x_prev, a: float, b: float, c: float, atol: float, rtol: float):
    """The closing launch of an adaptive order-2 step: (x_lower, x_higher, E) with E a one-element device tensor (include/gvf_dit.h)."""
    B = x.shape[0]
    n_per = x.numel() // max(B, 1)
    L = _lib.lib()
    scratch = torch.empty(int(L.gvf_dpm_err_scratch_doubles(B, n_per)), dtype=torch.float64, device=x.device)
    x_lower, x_higher = torch.empty_like(x), torch.empty_like(x)
    E = torch.empty(1, dtype=torch.float32, device=x.device)
    _lib.check(L.gvf_dpm_second_err(_p(x), _p(m), _p(m1), _p(x_prev), float(a), float(b), float(c), float(atol), float(rtol), _p(x_lower), _p(x_higher),
                                    B, n_per, _p(scratch), _p(E), _stream(x)), "gvf_dpm_second_err")
    return x_lower, x_higher, E
