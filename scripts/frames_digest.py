"""sha256 of the frames (and instance counts) of ONE bench step (BASELINE configs[1] workload) and of one live-shape chunk, for comparing builds of the
library bit for bit:  GVF_LIB=<variant .so> python scripts/frames_digest.py   (gvfdiffusion_amd._build --variant; scripts/gpu_ab.sh for timing)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w = bench.RasterWorkload(dev, 262_144, 800, 24, 2, seed=0)
    w.step()
    torch.cuda.synchronize()
    h = hashlib.sha256(w.color.cpu().numpy().tobytes()).hexdigest()
    print("lib", os.environ.get("GVF_LIB") or "product", "| bench step: frames sha256", h[:32], "instances", int(w.nr.to(torch.int64).sum()))
    from gvfdiffusion_amd import synthetic
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from gvfdiffusion_amd.utils import orbit_cameras, render_sample_frames
    attrs = synthetic.random_gaussians(262_144, sh_degree=0, seed=7)
    delta = (torch.randn((2, 262_144, 14), generator=torch.Generator().manual_seed(11)) * 0.01).to(dev)
    gm = synthetic.gaussian_model_from(attrs, 0, dev)
    rend = GaussianRenderer({"resolution": 512, "near": synthetic.NEAR, "far": synthetic.FAR, "ssaa": 1, "bg_color": (1, 1, 1)})
    rend.pipe.kernel_size = synthetic.KERNEL_2D
    K, cams = synthetic.intrinsics().to(dev), orbit_cameras(48).to(dev)
    hh = hashlib.sha256()
    with torch.no_grad():
        for _, frames in render_sample_frames(rend, gm, delta, K, extrinsics=cams, chunk_frames=96, streams=1):
            hh.update(frames.cpu().numpy().tobytes())
    print("lib", os.environ.get("GVF_LIB") or "product", "| live shape, 2 timesteps x 48 cameras: uint8 frames sha256", hh.hexdigest()[:32])


if __name__ == "__main__":
    main()
