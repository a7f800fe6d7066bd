"""In-tree build of the gfx950 HIP library (libgvf_hip.so).

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the CPU-only build
container; the resulting .so sits next to this file (git-ignored, but shipped to the GPU box).
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(_HERE, "libgvf_hip.so")

# -fno-slp-vectorize for EVERY source (round 4): left on, clang packs neighbouring scalar fp32 operations into v_pk_fma_f32 / v_pk_mul_f32 /
# v_pk_add_f32 -- and on gfx950 a wave's packed-fp32 results come out WRONG while another wave of the same CU issues MFMAs
# (scripts/ubench/coresident_victim.hip: the fp32 adaLN GEMV beside a pure-MFMA kernel, 218 of 512 launches wrong with v_pk_*, 0 of 512
# without; this was the "two samples in flight differ in their last bits" of rounds 3-4: modulation_f32_kernel of one sample's DiT step
# sharing CUs with the other sample's VAE-decode GEMMs).  No kernel of this library may contain packed fp32 arithmetic
# (tests/test_capi_symbols.py checks the code objects); the packed forms bought nothing beside MFMAs anyway (MI355X_MICROARCH.md).
# (-packed-fp32-ops: the target feature itself is switched off as well -- vector-typed source arithmetic, e.g. float4 += float4, selects the
# packed instructions without any vectoriser.)
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize",
          "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# per-source extra flags.  rast.hip: the floating-point contract shared with oracle/rast_oracle.c
# (no implicit fma contraction; fmaf only where written).
SOURCES = {
    "sort.hip": [],
    # -fno-slp-vectorize: left on, clang packs neighbouring scalar fp32 operations of the compositing loop into
    # v_pk_* instructions (4 cycles each against 2.8 for the scalar form, plus the v_mov traffic that builds the pairs)
    "rast.hip": ["-ffp-contract=off"],
    "vox2seq.hip": [],
    "resize.hip": [],
    # the squared distances must round exactly as the oracle's binary32 expression does (index-exact parity)
    "fps.hip": ["-ffp-contract=off"],
    # MFMA results straight into VGPRs (gfx950 has a unified register file): removes the accvgpr
    # read/write traffic between the MFMAs and the softmax / epilogue VALU code.
    # -fno-honor-nans: no canonicalising v_max in front of fmaxf (infinities stay honoured: -inf masks keys).
    # iterative-ilp: the scheduler variant that measured best for the attention kernels in the denoise step (8.5 -> 8.4 ms
    # per NFE; max-ilp and the default are slower; for rast.hip every non-default strategy slows the blend)
    "attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
    # tiled-cache cross attention: the issue order of its inner loop is written out (sched_barrier fences), so no scheduler flag
    "attn_xt.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
    "attn_xt64.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
    "gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
    "gemm256.hip": [],                       # accumulators in AGPRs: 256 of them per wave
    "gemm8.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
    # solver state updates: every operation rounded on its own (HIP's __fmul_rn / __fadd_rn are plain operators, so the contraction must be off)
    "dpm.hip": ["-ffp-contract=off"],
    # row-block kernel: default flags (accumulators in AGPRs: the kernel lives on the 512-register file of a 2-waves-per-SIMD launch)
    "rowblock.hip": [],
    "elem.hip": [],
    "vae.hip": [],
}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 HIP library cannot be built")


STAMP_PATH = LIB_PATH + ".srchash"


def source_hash() -> str:
    """sha256 over every file the library is built from (csrc/*.hip, csrc/*.h, include/*.h)."""
    import hashlib
    inc = os.path.join(os.path.dirname(_HERE), "include")
    files = sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] +
                   [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def raster_source_hash() -> str:
    """sha256 over the sources of the rasteriser kernels only: stamps profiles/*_pmc_raster.json so that bench.py can tell
    whether the committed HBM-traffic counters were taken on the kernels it is running."""
    import hashlib
    inc = os.path.join(os.path.dirname(_HERE), "include")
    h = hashlib.sha256()
    for f in (os.path.join(CSRC, "rast.hip"), os.path.join(CSRC, "sort.hip"), os.path.join(CSRC, "gvf_common.h"), os.path.join(inc, "gvf_rast.h")):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(os.path.dirname(_HERE), "include")
    headers += [os.path.join(inc, f) for f in os.listdir(inc)]
    objs = []
    for src, extra in SOURCES.items():
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        if force or _stale(obj, [sp, __file__] + headers):
            cmd = [hipcc] + COMMON + extra + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    with open(STAMP_PATH, "w") as f:       # lets _lib.lib() refuse a library that is older than its sources
        f.write(source_hash())
    return LIB_PATH


def build_variant(name: str, defines: dict, verbose: bool = False) -> str:
    """A/B aid: a second library `variants/libgvf_hip_<name>.so` in which the sources named in `defines` ({"attn_xt.hip": ["-DXT_RING_STAGES=4"],
    ...}) are compiled with extra flags and every other object is the product's.  Selected at run time with GVF_LIB=<path>
    (gvfdiffusion_amd/_lib.py), so ONE gpurun call can alternate variants on one box (scripts/gpu_ab.sh); never loaded by default."""
    build()
    vdir = os.path.join(OBJ_DIR, "variant_" + name)
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(os.path.join(_HERE, "variants"), exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for src, extra in SOURCES.items():
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        if src in defines:
            obj = os.path.join(vdir, src.replace(".hip", ".o"))
            cmd = [hipcc] + COMMON + extra + list(defines[src]) + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    out = os.path.join(_HERE, "variants", f"libgvf_hip_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) >= 3 and sys.argv[1] == "--variant":
        # python -m gvfdiffusion_amd._build --variant ring4 attn_xt.hip=-DXT_RING_STAGES=4 [rast.hip=-DFOO=1,-DBAR=2 ...]
        print(build_variant(sys.argv[2], {a.split("=", 1)[0]: a.split("=", 1)[1].split(",") for a in sys.argv[3:]}, verbose=True))
    else:
        print(build(force=False, verbose=True))
