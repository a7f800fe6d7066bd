"""Batched render driver (gvfdiffusion_amd/utils/inference_utils.py) = the reference's render_and_save_images loop
(utils/inference_utils.py:239-281) without the per-frame host round trips: same cameras, same (timestep, camera)
order, uint8 frames equal to the per-frame facade render followed by the reference's clamp -> *255 -> uint8."""
import os

import numpy as np
import pytest
import torch

from gvfdiffusion_amd import synthetic


def test_orbit_cameras_are_rigid_and_look_at_origin():
    from gvfdiffusion_amd.utils import orbit_cameras
    w2c = orbit_cameras(8).double()
    assert w2c.shape == (8, 4, 4)
    R, t = w2c[:, :3, :3], w2c[:, :3, 3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, dtype=torch.float64).expand(8, 3, 3), atol=1e-6)
    assert torch.allclose(torch.linalg.det(R), torch.ones(8, dtype=torch.float64), atol=1e-6)
    # the origin sits on the optical axis at distance 2 (COLMAP: +z forward), the world z axis is "up" (-y in the image)
    assert torch.allclose(t, torch.tensor([0.0, 0.0, 2.0], dtype=torch.float64).expand(8, 3), atol=1e-6)
    assert (R[:, 1, 2] < -0.99).all()
    eye = -(R.transpose(1, 2) @ t[:, :, None])[:, :, 0]
    assert torch.allclose(eye.norm(dim=1), torch.full((8,), 2.0, dtype=torch.float64), atol=1e-6) and eye[:, 2].abs().max() < 1e-6
    assert not torch.allclose(eye[0], eye[1])


def test_camera_blocks_are_cached_by_tensor_identity_and_version():
    """GaussianRenderer.make_frames keeps the camera blocks of the last few (extrinsics, intrinsics) pairs: the same tensors give
    the same list back, an in-place change of the cameras (or another delta mapping) rebuilds it.  Host logic only (no GPU)."""
    from gvfdiffusion_amd.renderers import GaussianRenderer
    rend = GaussianRenderer({"resolution": 64, "near": 0.8, "far": 1.6, "ssaa": 1, "bg_color": (1, 1, 1)})
    from gvfdiffusion_amd.utils import orbit_cameras
    ext = orbit_cameras(3).clone()
    K = torch.tensor([[1.2, 0.0, 0.5], [0.0, 1.2, 0.5], [0.0, 0.0, 1.0]])
    a = rend.make_frames(ext, K, [0, 1, 2])
    assert rend.make_frames(ext, K, [0, 1, 2]) is a
    assert rend.make_frames(ext, K, [0, 0, 0]) is not a and rend.make_frames(ext, K, [0, 0, 0])[2].delta_index == 0
    before = [a[1].viewmatrix[k] for k in range(16)]
    ext[1, 0, 3] += 0.25                                              # in place: same storage, new version
    b = rend.make_frames(ext, K, [0, 1, 2])
    assert b is not a and [b[1].viewmatrix[k] for k in range(16)] != before
    assert [b[0].viewmatrix[k] for k in range(16)] == [a[0].viewmatrix[k] for k in range(16)]
    fresh = GaussianRenderer({"resolution": 64, "near": 0.8, "far": 1.6, "ssaa": 1, "bg_color": (1, 1, 1)}).make_frames(ext, K, [0, 1, 2])
    assert all([b[f].projmatrix[k] for k in range(16)] == [fresh[f].projmatrix[k] for k in range(16)] for f in range(3))


def test_schedule_blocks_copied_from_the_orbit_equal_the_ones_built_per_chunk():
    """The chunked driver builds the orbit's camera blocks once and copies them per (timestep, camera) pair: byte for byte what
    `make_frames` builds from the gathered cameras with that delta mapping.  Host logic only (no GPU)."""
    import ctypes
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from gvfdiffusion_amd.utils import orbit_cameras
    rend = GaussianRenderer({"resolution": 64, "near": 0.8, "far": 1.6, "ssaa": 1, "bg_color": (1, 1, 1)})
    ext = orbit_cameras(5)
    K = torch.tensor([[1.2, 0.0, 0.5], [0.0, 1.2, 0.5], [0.0, 0.0, 1.0]])
    blocks = rend.make_frames(ext, K)
    assert all(b.delta_index == -1 for b in blocks)
    part = [(2, 4), (0, 1), (7, 1), (3, 0)]
    got = rend.frames_with_delta_index(blocks, part)
    want = rend.make_frames(ext[torch.tensor([c for _, c in part])], K, [t for t, _ in part])
    assert [bytes(g) for g in got] == [bytes(w) for w in want] and ctypes.sizeof(got[0]) == 160
    assert all(b.delta_index == -1 for b in blocks)                   # the orbit's blocks are templates: untouched

    class _Model:                                                     # render_frames refuses a block that selects a slice delta_pc lacks,
        _xyz = torch.zeros(4, 3)                                      # before any device work (the C entry point would return GVF_EINVAL)
    with pytest.raises(ValueError):
        rend.render_frames(_Model(), None, None, delta_pc=torch.zeros(3, 4, 14), frames=got)     # (7, 1) needs 8 slices
    with pytest.raises(ValueError):
        rend.render_frames(_Model(), None, None, delta_pc=None, frames=got[:1])                  # a delta index without deltas


@pytest.mark.gpu
def test_driver_matches_per_frame_renders(cuda):
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from gvfdiffusion_amd.utils import orbit_cameras, render_sample_frames
    P, T, V, S = 20_000, 3, 5, 160
    attrs = synthetic.random_gaussians(P, sh_degree=0, seed=7, scale_lo=0.004, scale_hi=0.02)
    gm = synthetic.gaussian_model_from(attrs, 0, cuda)
    delta = synthetic.random_deltas(T, P, seed=8, std=0.02).to(cuda)
    K = synthetic.intrinsics().to(cuda)
    rend = GaussianRenderer({"resolution": S, "near": synthetic.NEAR, "far": synthetic.FAR, "bg_color": (1, 1, 1)})
    rend.pipe.kernel_size = synthetic.KERNEL_2D
    cams = orbit_cameras(V)
    got, order = [], []
    for part, frames in render_sample_frames(rend, gm, delta, K, extrinsics=cams, chunk_frames=4):   # ragged last chunk
        assert frames.dtype == torch.uint8 and frames.shape[1:] == (3, S, S) and frames.is_cuda
        got.append(frames)
        order += part
    got = torch.cat(got)
    assert order == [(t, c) for t in range(T) for c in range(V)] and got.shape[0] == T * V
    assert rend.pipe.use_mip_gaussian is False                       # restored
    rend.pipe.use_mip_gaussian = True
    for k in (0, 4, 7, 14):
        t, c = order[k]
        with torch.no_grad():                                                     # as inference_dpm_latent.py:171 runs it
            one = rend.render(gm, cams[c].to(cuda), K, delta_pc=delta[t])["rgb"]     # the reference's per-frame call
        ref = (one.clamp(0.0, 1.0).cpu().numpy() * 255).astype("uint8")         # inference_utils.py:276-281
        d = np.abs(got[k].cpu().numpy().astype(int) - ref.astype(int))
        # torch activations (facade) vs fused activations (driver): sub-ulp colour differences can flip a uint8 step
        assert d.max() <= 1 and (d > 0).mean() < 0.02, (k, d.max(), (d > 0).mean())
    assert (got[0].float() - got[V].float()).abs().max() > 0          # the deltas move the object between timesteps
    # chunks in flight on 1, 2 or 3 streams (and a too-small first-chunk estimate: dense views later force the re-render path): same frames
    for n_streams in (1, 3):
        again = torch.cat([f for _, f in render_sample_frames(rend, gm, delta, K, extrinsics=cams, chunk_frames=4, streams=n_streams)])
        assert torch.equal(again, got), n_streams
    rend.pipe.use_mip_gaussian = False


@pytest.mark.gpu
def test_render_frames_supersamples_like_the_per_frame_render(cuda):
    """rendering_options.ssaa > 1 (renderers/gaussian_render.py:355-360: render at resolution x ssaa, bicubic antialiased down-sampling):
    the batched call == the reference's per-frame render() with the same option."""
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from gvfdiffusion_amd.utils import orbit_cameras
    P, T, S = 10_000, 3, 96
    attrs = synthetic.random_gaussians(P, sh_degree=1, seed=3, scale_lo=0.004, scale_hi=0.02)
    gm = synthetic.gaussian_model_from(attrs, 1, cuda)
    delta = synthetic.random_deltas(T, P, seed=4, std=0.02).to(cuda)
    K = synthetic.intrinsics().to(cuda)
    rend = GaussianRenderer({"resolution": S, "near": synthetic.NEAR, "far": synthetic.FAR, "bg_color": (1, 1, 1), "ssaa": 2})
    rend.pipe.use_mip_gaussian = True
    cams = orbit_cameras(T).to(cuda)
    with torch.no_grad():
        out = rend.render_frames(gm, cams, K, delta_pc=delta)
        assert out.rgb.shape == (T, 3, S, S)
        for t in range(T):
            one = rend.render(gm, cams[t], K, delta_pc=delta[t])["rgb"]
            assert one.shape == (3, S, S) and float((out.rgb[t] - one).abs().max()) < 2e-5
    rend.rendering_options.ssaa = 1
    with torch.no_grad():
        assert float((rend.render_frames(gm, cams, K, delta_pc=delta).rgb - out.rgb).abs().max()) > 1e-3     # the option does something


@pytest.mark.gpu
def test_all_delta_module_is_the_rgb_only_facade(cuda):
    """renderers/gaussian_render_all_delta.py (named by BASELINE.json's north_star): same frames as gaussian_render, rgb only."""
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from gvfdiffusion_amd.renderers import gaussian_render_all_delta as ad
    P, S = 5000, 96
    attrs = synthetic.random_gaussians(P, sh_degree=0, seed=9, scale_lo=0.004, scale_hi=0.02)
    gm = synthetic.gaussian_model_from(attrs, 0, cuda)
    delta = synthetic.random_deltas(1, P, seed=10, std=0.02).to(cuda)[0]
    K = synthetic.intrinsics().to(cuda)
    opts = {"resolution": S, "near": synthetic.NEAR, "far": synthetic.FAR, "bg_color": (1, 1, 1)}
    a, b = ad.GaussianRenderer(opts), GaussianRenderer(opts)
    for r in (a, b):
        r.pipe.use_mip_gaussian = True
        r.pipe.kernel_size = synthetic.KERNEL_2D
    ext = synthetic.orbit_w2c(40.0, 10.0).to(cuda)
    with torch.no_grad():
        ra, rb = a.render(gm, ext, K, delta_pc=delta), b.render(gm, ext, K, delta_pc=delta)
    assert set(ra.keys()) == {"rgb"} and torch.equal(ra["rgb"], rb["rgb"])


@pytest.mark.gpu
def test_render_and_save_images_writes_the_reference_files(cuda, tmp_path):
    """utils/inference_utils.py:208-297: file names, 512x512 frames, content = PIL's resize / pad of the per-frame render."""
    from types import SimpleNamespace
    from PIL import Image
    from gvfdiffusion_amd.attrdict import edict
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from gvfdiffusion_amd.utils import orbit_cameras, render_and_save_images
    P, T, V, S = 4000, 2, 3, 200
    attrs = synthetic.random_gaussians(P + 50, sh_degree=0, seed=11, scale_lo=0.004, scale_hi=0.02)
    full = synthetic.gaussian_model_from(attrs, 0, cuda)
    gm = synthetic.gaussian_model_from({k: v[:P] for k, v in attrs.items()}, 0, cuda)
    delta = synthetic.random_deltas(T, P + 50, seed=12, std=0.02).to(cuda)             # 50 padded rows past valid_idx
    rend = GaussianRenderer({"resolution": S, "near": synthetic.NEAR, "far": synthetic.FAR, "bg_color": (1, 1, 1)})
    rend.pipe.kernel_size = synthetic.KERNEL_2D
    K = synthetic.intrinsics()
    files = render_and_save_images(SimpleNamespace(exp_name=str(tmp_path)), SimpleNamespace(renderers=edict({"MipGS": rend})), [gm], delta[None],
                                   {"cams": {"intrinsics": K[None, None]}}, [P], 7, SimpleNamespace(device=cuda, process_index=1), [0.8],
                                   n_timesteps=T, n_views=V, chunk_frames=4)
    assert len(files) == T * V and os.path.basename(files[4]) == "rank_01_render_000007_cam_001_timesteps_01.png"
    rend.pipe.use_mip_gaussian = True
    cams = orbit_cameras(V)
    for path, (t, c) in ((files[0], (0, 0)), (files[4], (1, 1))):
        got = np.asarray(Image.open(path))
        assert got.shape == (512, 512, 3)
        with torch.no_grad():
            one = rend.render(gm, cams[c].to(cuda), K.to(cuda), delta_pc=delta[t, :P])["rgb"]
        rgb = (one.clamp(0.0, 1.0).permute(1, 2, 0).cpu().numpy() * 255).astype("uint8")
        img = Image.fromarray(rgb).resize((409, 409), resample=Image.Resampling.LANCZOS)       # int(512 * 0.8)
        want = Image.new("RGB", (512, 512), (255, 255, 255))
        want.paste(img, ((512 - 409) // 2, (512 - 409) // 2))
        d = np.abs(got.astype(int) - np.asarray(want).astype(int))
        assert d.max() <= 2 and (d > 0).mean() < 0.02, (d.max(), (d > 0).mean())       # fused vs torch activations: rare 1-step flips
    del full


def test_matrix_to_quaternion_round_trip():
    from gvfdiffusion_amd.utils.inference_utils import build_rotation, matrix_to_quaternion
    q = torch.randn(500, 4, generator=torch.Generator().manual_seed(0))
    R = build_rotation(q)
    q2 = matrix_to_quaternion(R)
    assert (build_rotation(q2) - R).abs().max() < 2e-6 and (q2[:, 0] >= 0).all() and (q2.norm(dim=1) - 1).abs().max() < 1e-6
    half = torch.tensor([[0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])              # 180-degree turns: real part 0
    assert (build_rotation(matrix_to_quaternion(build_rotation(half))) - build_rotation(half)).abs().max() < 1e-6


@pytest.mark.gpu
def test_align_gaussian_to_canonical_recovers_a_known_azimuth(cuda):
    """utils/inference_utils.py:38-178 without its CLIP term: an asymmetric object seen from azimuth 37 (zoomed by 1.2) as the
    canonical view -> best azimuth 37, scale factor ~1.2, and the turned model looks like the canonical view from the front."""
    from types import SimpleNamespace
    from gvfdiffusion_amd.attrdict import edict
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from gvfdiffusion_amd.utils.inference_utils import align_gaussian_to_canonical, azimuth_cameras, build_rotation
    P = 6000
    g = torch.Generator().manual_seed(5)
    attrs = synthetic.random_gaussians(P, sh_degree=0, seed=13, scale_lo=0.006, scale_hi=0.015)
    xyz = (torch.rand(P, 3, generator=g) - 0.5) * torch.tensor([0.5, 0.12, 0.35])         # a slab ...
    xyz[: P // 3] = (torch.rand(P // 3, 3, generator=g) - 0.5) * torch.tensor([0.1, 0.3, 0.1]) + torch.tensor([0.2, 0.2, 0.1])  # ... with a fin
    attrs["means3D"] = xyz
    attrs["shs"][: P // 3] = 1.5
    attrs["opacities"] = attrs["opacities"].clamp(min=0.6)
    gm = synthetic.gaussian_model_from(attrs, 0, cuda)
    rend = GaussianRenderer({"resolution": 512, "near": synthetic.NEAR, "far": synthetic.FAR, "bg_color": (1, 1, 1)})
    rend.pipe.kernel_size = synthetic.KERNEL_2D
    K = synthetic.intrinsics().to(cuda)
    Kz = K.clone(); Kz[0, 0] *= 1.2; Kz[1, 1] *= 1.2                                     # the canonical photo is closer
    with torch.no_grad():
        canon = rend.render_frames(gm, azimuth_cameras([37]).to(cuda), Kz, want_alpha_depth=True)
    xyz0, cov0 = gm.get_xyz.clone(), gm.get_covariance().clone()
    vae = SimpleNamespace(renderers=edict({"MipGS": rend}))
    model, scale = align_gaussian_to_canonical(gm, canon.rgb[0].clamp(0, 1), canon.alpha[0], K, vae, id=0, device=cuda, in_the_wild=True)
    assert model is gm and abs(scale - 1.2) < 0.04, scale
    a = np.radians(-37.0)
    R = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32, device=cuda)
    assert (gm.get_xyz - xyz0 @ R.T).abs().max() < 1e-5                                   # positions turned by -37 degrees about z
    S6 = lambda c: torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)
    want = R @ S6(cov0) @ R.T
    assert (S6(gm.get_covariance()) - want).abs().max() < 1e-6 + 1e-3 * float(want.abs().max())   # ... and so did the covariances
    with torch.no_grad():
        front = rend.render_frames(gm, azimuth_cameras([0]).to(cuda), Kz).rgb[0]
    mse = float(((front.clamp(0, 1) - canon.rgb[0].clamp(0, 1)) ** 2).mean())
    assert 10 * np.log10(1.0 / max(mse, 1e-12)) > 35.0
    # the coarse search (4 azimuths) picks the closest quarter turn
    gm2 = synthetic.gaussian_model_from(attrs, 0, cuda)
    with torch.no_grad():
        canon2 = rend.render_frames(gm2, azimuth_cameras([-90]).to(cuda), K, want_alpha_depth=True)
    xyz2 = gm2.get_xyz.clone()
    align_gaussian_to_canonical(gm2, canon2.rgb[0].clamp(0, 1), canon2.alpha[0], K, vae, id=1, device=cuda, in_the_wild=False)
    a = np.radians(90.0)
    R2 = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32, device=cuda)
    assert (gm2.get_xyz - xyz2 @ R2.T).abs().max() < 1e-5


@pytest.mark.gpu
def test_batched_rasteriser_call_is_capturable_in_a_graph(cuda):
    """include/gvf_rast.h: gvf_rast_forward_batched enqueues kernels only (camera blocks travel as kernel arguments, the per-call tables
    are cleared by the first launch, no host synchronisation), so a render loop can capture it once and replay it: frames of the replays
    equal the eager call's, also after the inputs changed in place."""
    import bench
    w = bench.RasterWorkload(cuda, 20_000, 256, 6, 2, seed=5)
    w.step()
    torch.cuda.synchronize()
    eager, eager_nr = w.color.clone(), w.nr.clone()
    assert int(eager_nr.sum()) > 0 and float(eager.std()) > 0
    s = torch.cuda.Stream(device=cuda)
    s.wait_stream(torch.cuda.current_stream(cuda))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        w.color.zero_()
        with torch.cuda.graph(g, stream=s):
            w.step()
    for _ in range(2):
        w.color.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(w.color, eager) and torch.equal(w.nr, eager_nr)
    w.delta.mul_(0.5)                                   # same buffers, new contents: the replay renders the new sample
    g.replay()
    torch.cuda.synchronize()
    replayed = w.color.clone()
    w.step()
    torch.cuda.synchronize()
    assert torch.equal(w.color, replayed) and not torch.equal(replayed, eager)
