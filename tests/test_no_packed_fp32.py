"""No kernel of libgvf_hip.so may contain packed-fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).

Round 4 root cause of "two samples in flight differ from the serial run": on gfx950 a wave's packed-fp32 results come out wrong while another
wave of the same CU issues MFMAs (scripts/ubench/coresident_victim.hip: a library-free reproducer; the fp32 adaLN GEMV of one sample's DiT step
beside the other sample's VAE-decode GEMMs).  gvfdiffusion_amd/_build.py therefore builds every source with -fno-slp-vectorize and the
packed-fp32-ops target feature off; this test disassembles the shipped code objects and holds the build to it."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def test_library_has_no_packed_fp32_arithmetic(tmp_path):
    lib = os.path.join(ROOT, "gvfdiffusion_amd", "libgvf_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not in this image")
    work = tmp_path / "lib.so"
    shutil.copy(lib, work)
    subprocess.run([OBJDUMP, "--offloading", str(work)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)     # writes lib.so.N.<target>
    objs = [f for f in os.listdir(tmp_path) if f.endswith("gfx950")]
    assert len(objs) >= 8, f"expected one gfx950 code object per source, found {objs}"
    bad = {}
    n_insts = 0
    for f in objs:
        asm = subprocess.run([OBJDUMP, "-d", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        n_insts += asm.count("\n")
        hits = re.findall(r"v_pk_(?:fma|mul|add)_f32", asm)
        if hits:
            bad[f] = len(hits)
    assert n_insts > 50_000, "the disassembly looks empty"
    assert not bad, f"packed fp32 arithmetic in the library's kernels: {bad}"
