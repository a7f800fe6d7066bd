"""Which 16-bit type the matrix pipe contracts (the `dtype` argument of include/gvf_dit.h): torch.float16 or torch.bfloat16.

The reference has no dtype argument either: its modules are fp32 and run under accelerate's autocast -- mixed_precision='fp16' when
--use_fp16 is given (inference_dpm_latent.py:122-125, README's command), i.e. fp16 operands with fp32 accumulation for every Linear and
attention, fp32 LayerNorm / softmax / RMSNorm.  BASELINE.json names bf16 for the MI355X build.  Both are built at the same MFMA rate and
the choice is made in the reference's own terms, first match wins:
  1. an explicit request: `module.set_compute_dtype(...)`,
  2. the tensors: 16-bit inputs of an operator are contracted in their own type (never silently down-cast),
  3. the GVF_DIT_DTYPE environment variable ("fp16" | "bf16"),
  4. an active torch.autocast region: its dtype (the accelerate path of the reference),
  5. the module's own default: DiT(use_fp16=True) -- configs/diffusion.yml -- means fp16, otherwise bf16."""
import os

import torch

LP_DTYPES = (torch.bfloat16, torch.float16)
_NAMES = {"fp16": torch.float16, "float16": torch.float16, "half": torch.float16, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}


def parse(name):
    if name is None or isinstance(name, torch.dtype):
        if name is not None and name not in LP_DTYPES:
            raise ValueError(f"compute dtype must be torch.float16 or torch.bfloat16, got {name}")
        return name
    try:
        return _NAMES[str(name).lower()]
    except KeyError:
        raise ValueError(f"compute dtype must be one of {sorted(_NAMES)}, got {name!r}") from None


def from_env():
    return parse(os.environ.get("GVF_DIT_DTYPE"))


def autocast_dtype():
    """dtype of the active torch.autocast region for the GPU, or None."""
    try:
        on = torch.is_autocast_enabled("cuda")
        dt = torch.get_autocast_dtype("cuda") if on else None
    except TypeError:                      # older signature
        on = torch.is_autocast_enabled()
        dt = torch.get_autocast_gpu_dtype() if on else None
    return dt if dt in LP_DTYPES else None


def resolve(explicit=None, tensors=(), default=torch.bfloat16):
    if explicit is not None:
        return explicit
    for t in tensors:
        if t is not None and t.dtype in LP_DTYPES:
            return t.dtype
    return from_env() or autocast_dtype() or default
