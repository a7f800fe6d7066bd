"""ctypes binding of the C ABI in include/*.h (libgvf_hip.so).

There is no CPU fallback anywhere in this package: if the HIP library is missing, every operator
raises.  PyTorch is plumbing here (device memory, streams); the compute is in the library.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgvf_hip.so")

GVF_OK, GVF_EINVAL, GVF_ENOSPC, GVF_ELAUNCH = 0, -1, -2, -3
_ERR = {GVF_EINVAL: "GVF_EINVAL (bad argument)", GVF_ENOSPC: "GVF_ENOSPC (workspace too small)",
        GVF_ELAUNCH: "GVF_ELAUNCH (HIP launch/runtime failure)"}

RAST_MODE_MIP, RAST_MODE_DILATE = 0, 1
RAST_BIN_AUTO, RAST_BIN_RADIX, RAST_BIN_BUCKET = 0, 1, 2


class GvfError(RuntimeError):
    pass


class GvfRastFrame(ctypes.Structure):
    _fields_ = [("viewmatrix", ctypes.c_float * 16), ("projmatrix", ctypes.c_float * 16),
                ("campos", ctypes.c_float * 3), ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float),
                ("delta_index", ctypes.c_int32), ("reserved", ctypes.c_int32 * 2)]


class GvfRastSettings(ctypes.Structure):
    _fields_ = [("image_height", ctypes.c_int32), ("image_width", ctypes.c_int32), ("sh_degree", ctypes.c_int32),
                ("mode", ctypes.c_int32), ("kernel_size", ctypes.c_float), ("scale_modifier", ctypes.c_float),
                ("bg", ctypes.c_float * 3), ("prefiltered", ctypes.c_int32), ("debug", ctypes.c_int32),
                ("upstream_binning", ctypes.c_int32), ("bin_algo", ctypes.c_int32)]


class GvfGaussianActivation(ctypes.Structure):
    _fields_ = [("aabb", ctypes.c_float * 6), ("scale_bias", ctypes.c_float), ("opacity_bias", ctypes.c_float),
                ("min_kernel_size", ctypes.c_float), ("scaling_activation", ctypes.c_int32)]


_vp, _i, _i64, _sz, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float

# name -> (restype, argtypes); mirrors include/*.h one to one (checked by tests/test_capi_symbols.py)
SIGNATURES = {
    "gvf_version": (ctypes.c_char_p, []),
    "gvf_rast_workspace_bytes": (_i, [_i, _i, _i, _i, _i64, ctypes.POINTER(_sz)]),
    "gvf_rast_forward": (_i, [ctypes.POINTER(GvfRastSettings), ctypes.POINTER(GvfRastFrame), _i, _i,
                              _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gvf_rast_forward_batched": (_i, [ctypes.POINTER(GvfRastSettings), ctypes.POINTER(GvfRastFrame), _i,
                                      ctypes.POINTER(GvfGaussianActivation), _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _i, _vp, _sz, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gvf_rast_forward_batched_u8": (_i, [ctypes.POINTER(GvfRastSettings), ctypes.POINTER(GvfRastFrame), _i,
                                         ctypes.POINTER(GvfGaussianActivation), _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                         _i, _vp, _sz, _i64, _vp, _vp, _vp]),
    "gvf_rast_backward_scratch_bytes": (_i, [_i, ctypes.POINTER(_sz)]),
    "gvf_rast_backward": (_i, [ctypes.POINTER(GvfRastSettings), ctypes.POINTER(GvfRastFrame), _i, _i,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i64, _vp, _vp, _vp, _vp, _sz,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gvf_gaussian_activate": (_i, [ctypes.POINTER(GvfGaussianActivation), _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _vp, _vp, _vp]),
    "gvf_rgb_to_u8": (_i, [_vp, _vp, _i64, _vp]),
    "gvf_rast_sort_class_counts": (_i, [_vp, _sz, _i, _i, _i, _i, _i64, ctypes.POINTER(ctypes.c_uint32), _vp]),
    "gvf_rast_profile_enable": (_i, [_i]),
    "gvf_rast_profile_read": (_i, [ctypes.POINTER(_f), ctypes.POINTER(_i)]),
    "gvf_rast_shared_activation_calls": (_i64, []),
    "gvf_sort_tmp_bytes": (_sz, [_i64]),
    "gvf_sort_pairs_u64": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp, _sz, _vp]),
    "gvf_tile_sort_u64": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
}

_LIB = None


def register(signatures):
    """Other modules of the package (attention, gemm ...) add their C-ABI entry points here."""
    SIGNATURES.update(signatures)
    if _LIB is not None:
        _bind(_LIB, signatures)


def _bind(lib, signatures):
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GvfError(f"{LIB_PATH} is missing: build it with `python -m gvfdiffusion_amd._build` "
                           "(hipcc, gfx950). There is no CPU fallback.")
        # a library older than its sources silently runs yesterday's kernels: refuse (content hash written by _build)
        stamp = LIB_PATH + ".srchash"
        if os.path.exists(stamp) and os.environ.get("GVF_ALLOW_STALE_LIB") != "1":
            from . import _build
            if open(stamp).read().strip() != _build.source_hash():
                raise GvfError(f"{LIB_PATH} was built from different sources than the ones in csrc/ and include/: "
                               "rebuild with `python -m gvfdiffusion_amd._build`")
        # torch must be imported before the dlopen: its wheel bundles libamdhip64.so (SONAME
        # libamdhip64.so.7) and libgvf_hip.so must bind to THAT runtime instance to share torch's
        # device context and streams.  Loaded the other way round, the process ends up with two HIP
        # runtimes and this library's one reports "no ROCm-capable device".
        import torch  # noqa: F401
        # GVF_LIB=<path>: a variant build of the same sources with other compile-time switches (gvfdiffusion_amd._build.build_variant), for
        # A/B measurements on one box; it must export the same C ABI (bound below: a missing symbol raises)
        l = ctypes.CDLL(os.environ.get("GVF_LIB") or LIB_PATH)
        _bind(l, SIGNATURES)
        _LIB = l
    return _LIB


def check(rc: int, what: str):
    if rc != GVF_OK:
        raise GvfError(f"{what} failed: {_ERR.get(rc, rc)}")


def ptr(t):
    """data pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GvfError("gvfdiffusion_amd operators run on the MI355X only: got a CPU tensor "
                           "(there is no CPU fallback; the CPU oracle lives in oracle/ for tests)")
