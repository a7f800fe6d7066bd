"""How much of the per-tile sort's output does the blend look at?    Needs the BLEND_CONSUMED variant library:
    python -m gvfdiffusion_amd._build --variant blendc rast.hip=-DBLEND_CONSUMED
    GVF_LIB=gvfdiffusion_amd/variants/libgvf_hip_blendc.so python scripts/blend_consumed.py [live|bench]
Per size class of a (frame, tile) segment: segments, keys sorted, keys staged by the compositing loop when every pixel of the tile was saturated
(whole 256-key rounds), and the ratio -- for the reference's live render shape (one 96-frame chunk: 512^2, SH 0, 262 144 Gaussians) or the bench shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gvfdiffusion_amd import _lib, synthetic

dev = torch.device("cuda:0")
L = _lib.lib()
fn = L.gvf_debug_blend_consumed
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
which = sys.argv[1] if len(sys.argv) > 1 else "live"
assert fn(None, 1) == 0
if which == "bench":
    w = bench.RasterWorkload(dev, 262_144, 800, 24, 2, seed=0)
    assert fn(None, 1) == 0
    w.step(); torch.cuda.synchronize()
else:
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from gvfdiffusion_amd.utils import orbit_cameras, render_sample_frames
    P, T, V, S = 262_144, 2, 128, 512
    attrs = synthetic.random_gaussians(P, sh_degree=0, seed=7)
    delta = (torch.randn((T, P, 14), generator=torch.Generator().manual_seed(11)) * 0.01).to(dev)
    gm = synthetic.gaussian_model_from(attrs, 0, dev)
    rend = GaussianRenderer({"resolution": S, "near": synthetic.NEAR, "far": synthetic.FAR, "ssaa": 1, "bg_color": (1, 1, 1)})
    rend.pipe.kernel_size = synthetic.KERNEL_2D
    K, cams = synthetic.intrinsics().to(dev), orbit_cameras(V).to(dev)
    with torch.no_grad():
        n = sum(fr.shape[0] for _, fr in render_sample_frames(rend, gm, delta, K, extrinsics=cams, chunk_frames=96, streams=1))
        assert fn(None, 1) == 0
        n = sum(fr.shape[0] for _, fr in render_sample_frames(rend, gm, delta, K, extrinsics=cams, chunk_frames=96, streams=1))
    torch.cuda.synchronize()
    print(f"{which}: {n} frames of {S}x{S}")
out = (ctypes.c_ulonglong * 20)()
assert fn(out, 0) == 0
tot_s = tot_u = 0
for c, name in enumerate(("<= 2048 keys", "2049-4096", "4097-16384", "> 16384")):
    segs, srt, used, m0, bits = (out[5 * c + k] for k in range(5))
    tot_s += srt; tot_u += used
    print(f"{name:14s} segments {segs:9d}  keys sorted {srt:12d}  keys staged before saturation {used:12d}  ratio {used / max(srt, 1):.3f}"
          f"   of the staged: touching NO 8x8 quadrant of their tile {m0 / max(used, 1):.3f}, quadrants touched per key {bits / max(used, 1):.2f} of 4")
print(f"all            keys sorted {tot_s}  staged {tot_u}  ratio {tot_u / max(tot_s, 1):.3f}")
