"""Attention operator seam of the reference (model/attention/__init__.py:3-32): a module-global
BACKEND chosen by the ATTN_BACKEND environment variable at import and by set_backend() at run time.
The reference accepts xformers / flash_attn / sdpa / naive; this package provides one backend, `hip`
(csrc/attn.hip).  Any other value is refused loudly instead of silently falling back."""
import os
from typing import *

BACKEND = "hip"
DEBUG = False


def __from_env():
    global BACKEND, DEBUG
    env_attn_backend = os.environ.get("ATTN_BACKEND")
    env_attn_debug = os.environ.get("ATTN_DEBUG")
    if env_attn_backend is not None:
        set_backend(env_attn_backend)
    if env_attn_debug is not None:
        DEBUG = env_attn_debug == "1"


def set_backend(backend: str):
    global BACKEND
    if backend != "hip":
        raise ValueError(f"attention backend {backend!r} is not available in gvfdiffusion_amd: the only backend is "
                         "'hip' (hand-written gfx950 kernel); the reference's xformers/flash_attn/sdpa/naive "
                         "backends are CUDA/torch paths this package deliberately does not carry")
    BACKEND = backend


def set_debug(debug: bool):
    global DEBUG
    DEBUG = debug


__from_env()

from .full_attn import *  # noqa: E402,F401,F403
from .modules import *  # noqa: E402,F401,F403
