"""Disk cache for CPU-oracle results inside the test suite (the GPU suite's wall time is mostly the torch oracles of
the two VAEs at their released widths, evaluated several times on the same inputs -- once per operand type of the device run, and once more in
the child process of test_kv_resident_attention_variant_forced_everywhere).

A result is keyed by the sha1 of a tag, the oracle's source file, the configuration and the BYTES of every input tensor, so a hit is the same
computation on the same data by the same oracle code; nothing about the device path is cached.  Files live under $GVF_TEST_CACHE (default
<tmp>/gvf_oracle_cache), are shared by the processes of one run and harmless across runs.  GVF_TEST_CACHE=off disables the cache."""
import hashlib
import json
import os
import tempfile

import torch

_DIR = os.environ.get("GVF_TEST_CACHE", os.path.join(tempfile.gettempdir(), "gvf_oracle_cache"))


def _feed(h, obj):
    if torch.is_tensor(obj):
        t = obj.detach().cpu().contiguous()
        h.update(str((tuple(t.shape), str(t.dtype))).encode())
        h.update(t.view(torch.uint8).numpy().tobytes() if t.numel() else b"")
    elif isinstance(obj, dict):
        for k in sorted(obj):
            h.update(str(k).encode())
            _feed(h, obj[k])
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _feed(h, o)
    else:
        h.update(json.dumps(obj, sort_keys=True, default=str).encode())


def oracle_cached(tag, source_module, inputs, compute):
    """compute() -> a tensor or a (nested) tuple / list / dict of tensors; `inputs`: everything the result depends on (config, state dict,
    tensors, precision name); `source_module`: the oracle module whose source participates in the key."""
    if _DIR == "off":
        return compute()
    h = hashlib.sha1()
    h.update(tag.encode())
    h.update(open(source_module.__file__, "rb").read())
    _feed(h, inputs)
    path = os.path.join(_DIR, f"{tag}_{h.hexdigest()}.pt")
    if os.path.exists(path):
        try:
            return torch.load(path, map_location="cpu")
        except Exception:          # noqa: BLE001 -- a torn file of a concurrent writer: recompute
            pass
    out = compute()
    os.makedirs(_DIR, exist_ok=True)
    tmp = f"{path}.{os.getpid()}.tmp"
    torch.save(out, tmp)
    os.replace(tmp, path)
    return out
