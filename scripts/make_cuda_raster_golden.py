"""Emit tests/golden/raster_cuda_golden.npz from the REAL CUDA rasterisers -- run this on any CUDA box.

The reference's rasteriser arithmetic lives in two third-party CUDA packages that are neither vendored nor pinned
(/root/reference/setup.sh:111,220-227):
    diff_gaussian_rasterization  -- github.com/autonomousvision/mip-splatting, submodules/diff-gaussian-rasterization
    diff_gauss                   -- github.com/slothfulxtx/diff-gaussian-rasterization
They cannot be built in the MI355X build container (no nvcc, no network), so oracle/rast_oracle.c is "parity unpinned"
at pixel level.  This script closes that gap for whoever owns a CUDA GPU: it renders the exact scenes of
tests/test_rast_gpu.py::test_frame_matches_oracle (seeded through gvfdiffusion_amd/synthetic.py, which is pure Python /
torch and needs no HIP library) and frame 1 of BASELINE configs[1] with the real wheels, through the same call the
reference makes (renderers/gaussian_render.py:106-143,198-220), and stores colour / radii (/ depth, alpha).
tests/test_oracle_rast.py::test_oracle_matches_cuda_golden and tests/test_rast_gpu.py::test_hip_matches_cuda_golden
switch on as soon as the file exists (1e-3 max-abs on pixels, radii exact).

    pip install <the two packages as setup.sh does>;  python scripts/make_cuda_raster_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gvfdiffusion_amd import synthetic          # noqa: E402  (pure torch)
from rast_util import camera_block              # noqa: E402

# the scenes: (name, P, sh_degree, seed, scale_lo, scale_hi, H, W, azimuth, elevation) -- keep in sync with
# tests/test_rast_gpu.py::test_frame_matches_oracle and bench.py's configs[1] workload
SCENES = [(f"deg{deg}_{H}x{W}", 20_000, deg, 11 + deg, 0.003, 0.03, H, W, 40.0 * deg + 5, 12.0)
          for deg in range(4) for (H, W) in ((256, 256), (200, 312))]
SCENES.append(("config1_frame1", 262_144, 2, 0, 0.002, 0.01, 800, 800, 15.0, 0.0))


def package_info(modname):
    try:
        from importlib import metadata
        for dist in metadata.distributions():
            top = (dist.read_text("top_level.txt") or "").split()
            if modname in top or dist.metadata["Name"].replace("-", "_") == modname:
                return {"name": dist.metadata["Name"], "version": dist.version, "direct_url": dist.read_text("direct_url.json")}
    except Exception as e:              # noqa: BLE001
        return {"error": repr(e)}
    return {}


def main():
    assert torch.cuda.is_available(), "needs a CUDA device and the two CUDA rasteriser packages"
    dev = torch.device("cuda:0")
    import diff_gaussian_rasterization as mip     # mip-splatting fork: settings take kernel_size + subpixel_offset
    import diff_gauss as dg
    out = {"scenes_json": np.frombuffer(json.dumps(SCENES).encode(), dtype=np.uint8),
           "packages_json": np.frombuffer(json.dumps({"diff_gaussian_rasterization": package_info("diff_gaussian_rasterization"),
                                                      "diff_gauss": package_info("diff_gauss"),
                                                      "torch": torch.__version__, "cuda": torch.version.cuda}).encode(), dtype=np.uint8)}
    for name, P, deg, seed, slo, shi, H, W, azi, elev in SCENES:
        a = {k: v.to(dev) for k, v in synthetic.random_gaussians(P, sh_degree=deg, seed=seed, scale_lo=slo, scale_hi=shi).items()}
        cam = camera_block(azi=azi, elev=elev)
        common = dict(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                      bg=torch.tensor(synthetic.BG, device=dev), scale_modifier=1.0, viewmatrix=cam["viewmatrix"].to(dev),
                      projmatrix=cam["projmatrix"].to(dev), sh_degree=deg, campos=cam["campos"].to(dev), prefiltered=False, debug=False)
        kw = dict(means3D=a["means3D"], means2D=torch.zeros_like(a["means3D"]), shs=a["shs"], colors_precomp=None,
                  opacities=a["opacities"], scales=a["scales"], rotations=a["rotations"], cov3D_precomp=None)
        with torch.no_grad():
            st = mip.GaussianRasterizationSettings(kernel_size=synthetic.KERNEL_2D,
                                                   subpixel_offset=torch.zeros((H, W, 2), device=dev), **common)
            color, radii = mip.GaussianRasterizer(raster_settings=st)(**kw)
            ret = dg.GaussianRasterizer(raster_settings=dg.GaussianRasterizationSettings(**common))(**kw)
        crop = (slice(None), slice(272, 528), slice(272, 528)) if name.startswith("config1") else (slice(None),) * 3
        out[f"{name}.mip.color"] = color[crop].float().cpu().numpy()
        out[f"{name}.mip.radii"] = radii.int().cpu().numpy()
        c2, depth, _normal, alpha, r2 = ret[0], ret[1], ret[2], ret[3], ret[4]
        out[f"{name}.dilate.color"] = c2[crop].float().cpu().numpy()
        out[f"{name}.dilate.depth"] = depth[crop].float().cpu().numpy()
        out[f"{name}.dilate.alpha"] = alpha[crop].float().cpu().numpy()
        out[f"{name}.dilate.radii"] = r2.int().cpu().numpy()
        print(name, "mip visible", int((radii > 0).sum()), "dilate visible", int((r2 > 0).sum()))
    path = os.path.join(ROOT, "tests", "golden", "raster_cuda_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
