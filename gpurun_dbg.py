import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch, numpy as np
from gvfdiffusion_amd import synthetic, rasterizer as R
from gvfdiffusion_amd.renderers import GaussianRenderer
from rast_util import camera_block
cuda = torch.device('cuda:0')
P, deg, S, T = 30_000, 2, 208, 3
attrs = synthetic.random_gaussians(P, sh_degree=deg, seed=31, scale_lo=0.003, scale_hi=0.02)
gm = synthetic.gaussian_model_from(attrs, deg, cuda)
delta = synthetic.random_deltas(T, P, seed=5).to(cuda)
rend = GaussianRenderer({"resolution": S, "near": synthetic.NEAR, "far": synthetic.FAR, "bg_color": synthetic.BG})
rend.pipe.use_mip_gaussian = True
cams = [camera_block(azi=15.0 * f, elev=5.0) for f in range(4)]
ext = torch.stack([c["extrinsics"] for c in cams]).to(cuda)
K = cams[0]["intrinsics"].to(cuda)
idx = [0, 1, 2, -1]
from gvfdiffusion_amd import _lib
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 0
st = R.make_settings(S, S, deg, MODE, 0.1, 1.0, synthetic.BG)
frames = [R.make_frame(c["viewmatrix"], c["projmatrix"], c["campos"], c["tanfovx"], c["tanfovy"], idx[f]) for f, c in enumerate(cams)]
out = R.rasterize_batched(st, frames, gm.activation_struct(), gm._xyz, gm.get_features, gm._scaling, gm._rotation, gm._opacity, delta=delta, want_alpha_depth=True, want_radii=True)
for f in range(4):
    d = None if idx[f] < 0 else delta[idx[f]]
    act = R.gaussian_activate(gm.activation_struct(), gm._xyz, gm.get_features, gm._scaling, gm._rotation, gm._opacity, d)
    two = R.rasterize(st, frames[f], act["means3D"], act["opacities"], shs=act["shs"], scales=act["scales"], rotations=act["rotations"], want_alpha_depth=True)
    dif = (out["color"][f] - two["color"]).abs()
    print(f, "nr", int(out["num_rendered"][f]), two["num_rendered"], "radii equal", torch.equal(out["radii"][f], two["radii"]),
          "ndiff radii", int((out["radii"][f] != two["radii"]).sum()), "color maxdiff", float(dif.max()), "npix", int((dif.amax(0) > 0).sum()))
    # single frame batched (F=1) fused
    one = R.rasterize_batched(st, [frames[f]], gm.activation_struct(), gm._xyz, gm.get_features, gm._scaling, gm._rotation, gm._opacity, delta=delta, want_alpha_depth=True, want_radii=True)
    print("   F=1 fused vs F=4 fused:", float((one["color"][0] - out["color"][f]).abs().max()), " F=1 fused vs two-step:", float((one["color"][0] - two["color"]).abs().max()))
    two2 = R.rasterize(st, frames[f], act["means3D"], act["opacities"], shs=act["shs"], scales=act["scales"], rotations=act["rotations"], want_alpha_depth=True)
    print("   two-step repeat determinism:", float((two2["color"] - two["color"]).abs().max()))
