"""The C-ABI library loads without a GPU and exports every entry point include/*.h declares; the
ctypes signatures in gvfdiffusion_amd/_lib.py cover exactly that set (no compute calls here)."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        names |= set(re.findall(r"\b(gvf_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol():
    from gvfdiffusion_amd import _build, _lib
    import gvfdiffusion_amd.ops  # noqa: F401  (registers the remaining entry points)
    import gvfdiffusion_amd.utils  # noqa: F401  (gvf_fps)
    assert os.path.exists(_build.LIB_PATH), "libgvf_hip.so not built (python -m gvfdiffusion_amd._build)"
    lib = ctypes.CDLL(_build.LIB_PATH)
    declared = declared_functions()
    assert len(declared) >= 7
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert set(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ declared
    assert _lib.lib().gvf_version().decode().endswith("gfx950")


def test_host_side_argument_errors_need_no_gpu():
    from gvfdiffusion_amd import _lib
    l = _lib.lib()
    out = ctypes.c_size_t(0)
    assert l.gvf_rast_workspace_bytes(262144, 24, 800, 800, 30_000_000, ctypes.byref(out)) == 0
    assert out.value > 24 * 262144 * 48
    assert l.gvf_rast_workspace_bytes(-1, 1, 8, 8, 0, ctypes.byref(out)) == _lib.GVF_EINVAL
    assert l.gvf_rast_workspace_bytes(1, 0, 8, 8, 0, ctypes.byref(out)) == _lib.GVF_EINVAL
    assert l.gvf_sort_tmp_bytes(1 << 20) >= 256 * 256 * 4


def test_rowblock_argument_struct_layout_matches_the_library():
    """The ctypes mirror of gvf_rowblock_args has the size and field offsets the compiled library reports, and a null / malformed argument
    block is refused on the host (no GPU needed)."""
    from gvfdiffusion_amd import _lib
    from gvfdiffusion_amd.ops import dit_ops
    l = _lib.lib()
    buf = (ctypes.c_int32 * 32)()
    n = l.gvf_rowblock_args_layout(buf, 32)
    A = dit_ops.RowblockArgs
    mine = [ctypes.sizeof(A)] + [getattr(A, f).offset for f in ("x", "in_x", "gate1", "mod_ld", "b_fc1", "ln2", "b3", "hb_out", "k_tiles", "gamma_k", "kv_group_rows", "dtype",
                                                                       "t_frames", "t_b_qkv", "t_scale", "t_ln")]
    assert n == len(mine) and list(buf[:n]) == mine
    assert l.gvf_rowblock_fused_bf16(None, None) == _lib.GVF_EINVAL
    a = A()
    a.M, a.C, a.K1, a.lda = 96, 256, 128, 128                   # C != 512
    assert l.gvf_rowblock_fused_bf16(ctypes.byref(a), None) == _lib.GVF_EINVAL
    a.C, a.M = 512, 100                                          # rows not a multiple of 48
    assert l.gvf_rowblock_fused_bf16(ctypes.byref(a), None) == _lib.GVF_EINVAL
    a.M, a.dtype = 96, 7                                         # not a 16-bit operand type
    assert l.gvf_rowblock_fused(ctypes.byref(a), None) == _lib.GVF_EINVAL
    assert l.gvf_rowblock_packed_bytes(512, 64) == 512 * 128 * 2 and l.gvf_rowblock_packed_bytes(500, 64) == _lib.GVF_EINVAL


def test_operators_refuse_cpu_tensors():
    import torch
    from gvfdiffusion_amd import _lib, rasterizer as R
    st = R.make_settings(16, 16, 0, 0, 0.1, 1.0, (1, 1, 1))
    eye = torch.eye(4)
    fr = R.make_frame(eye, eye, torch.zeros(3), 0.5, 0.5)
    with pytest.raises(_lib.GvfError):
        R.rasterize(st, fr, torch.zeros(4, 3), torch.ones(4, 1), shs=torch.zeros(4, 1, 3), scales=torch.ones(4, 3),
                    rotations=torch.ones(4, 4))


def test_scores_bounded_hint_and_worst_case(monkeypatch):
    """dit_ops.scores_bounded (host logic of GVF_ATTN_SCORES_BOUNDED): 32 |gq gk| / sqrt(32) * log2 e * 1.01 = 8.24 |gq gk| against 15.5 octaves;
    default = root mean square of the gain product over a head's channels (a hint: the kernel's range guard is the guarantee), strict = the
    largest product of one channel (the worst case over every query / key direction)."""
    import torch
    from gvfdiffusion_amd.ops import dit_ops
    one = torch.ones((4, 32))
    assert dit_ops.scores_bounded(one, one) and dit_ops.scores_bounded(one, one, strict=True)
    assert dit_ops.scores_bounded(1.3 * one, 1.4 * one, strict=True) and not dit_ops.scores_bounded(1.4 * one, 1.4 * one)
    spiky = one.clone(); spiky[2, 5] = 2.5                     # one channel of one head with a large gain: typical directions still fit
    assert dit_ops.scores_bounded(spiky, one) and not dit_ops.scores_bounded(spiky, one, strict=True)
    loud_head = one.clone(); loud_head[1] = 2.0                # a whole head with gain 2: its scores reach 16.5 octaves
    assert not dit_ops.scores_bounded(loud_head, one)
    assert not dit_ops.scores_bounded(None, one) and not dit_ops.scores_bounded(one, None)      # no RMSNorm, no bound
    assert dit_ops.scores_bounded(one.reshape(2, 64), one.reshape(2, 64), head_dim=64)
    monkeypatch.setenv("GVF_ATTN_BOUNDED", "0")
    assert not dit_ops.scores_bounded(one, one)
