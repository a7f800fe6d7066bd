"""Forward + backward of the rasteriser operator at the BASELINE shape (262 144 Gaussians, 800 x 800, SH degree 2),
one frame per call as a training step sees it (train_vae.py:321-352); GPU only.  Prints ms per stage."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gvfdiffusion_amd import synthetic
from rast_util import camera_block
from test_rast_gpu import _settings

dev = torch.device("cuda:0")
P, S, deg = int(os.environ.get("GVF_P", 262144)), int(os.environ.get("GVF_S", 800)), 2
a = synthetic.random_gaussians(P, sh_degree=deg, seed=0, scale_lo=0.002, scale_hi=0.01)
leaves = {k: v.to(dev).requires_grad_(True) for k, v in a.items()}
m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
w = torch.randn((3, S, S), device=dev)
cams = [camera_block(azi=15.0 * f) for f in range(4)]

def step(cam, backward=True):
    rast = _settings(cam, S, S, deg, 0, dev)
    color, radii = rast(means3D=leaves["means3D"], means2D=m2, shs=leaves["shs"], colors_precomp=None, opacities=leaves["opacities"],
                        scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
    if backward:
        (color * w).sum().backward()

def timed(fn, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for c in cams: step(c)
n = 8
t_fb = timed(lambda i: step(cams[i % 4]), n)
t_f = timed(lambda i: step(cams[i % 4], False), n)
with torch.no_grad():
    t_inf = timed(lambda i: step(cams[i % 4], False), n)
print(f"P={P} {S}x{S} SH{deg}: forward+backward {t_fb:.2f} ms/frame, differentiable forward alone {t_f:.2f}, "
      f"no-grad forward {t_inf:.2f} (one frame per call, incl. host sync on num_rendered) => backward ~{t_fb - t_f:.2f} ms")
