"""Shared helpers of the rasteriser tests: camera blocks as the reference builds them
(renderers/gaussian_render.py:285-321), oracle invocation, flagged-pixel comparison."""
import numpy as np
import torch

from gvfdiffusion_amd import synthetic


camera_block = synthetic.camera_block        # (moved into the package: bench.py and the smoke test use it too)


def oracle_render(oracle, attrs, cam, H, W, sh_degree, mode=0, kernel_size=synthetic.KERNEL_2D, bg=synthetic.BG,
                  scale_modifier=1.0, colors_precomp=None, cov3D_precomp=None, subpixel_offset=None, brute=False, tight=False):
    n = lambda t: None if t is None else t.detach().cpu().numpy()
    use_cov = cov3D_precomp is not None
    return oracle.rast_render(
        n(attrs["means3D"]), None if colors_precomp is not None else n(attrs["shs"]), n(colors_precomp),
        n(attrs["opacities"]), None if use_cov else n(attrs["scales"]), None if use_cov else n(attrs["rotations"]),
        n(cov3D_precomp), H=H, W=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], kernel_size=kernel_size,
        scale_modifier=scale_modifier, mode=mode, viewmatrix=n(cam["viewmatrix"]), projmatrix=n(cam["projmatrix"]),
        campos=n(cam["campos"]), sh_degree=sh_degree, bg=np.asarray(bg, np.float32),
        subpixel_offset=n(subpixel_offset), brute=brute, tight=tight)


_SESSION_CACHE = {}


def cached(key, make):
    """Session cache for the oracle's full-size results (~1 s of CPU per 262 144-Gaussian frame, ~0.3 s per activation): the rasteriser tests
    run once per binning algorithm and ask for the same oracle frames both times (the GPU suite's wall time)."""
    if key not in _SESSION_CACHE:
        _SESSION_CACHE[key] = make()
    return _SESSION_CACHE[key]


def cam_from_frame(fr):
    """The camera dict oracle_render() takes, read back from a GvfRastFrame camera block (what the batched driver hands the kernels)."""
    return {"viewmatrix": torch.tensor(list(fr.viewmatrix), dtype=torch.float32).reshape(4, 4),
            "projmatrix": torch.tensor(list(fr.projmatrix), dtype=torch.float32).reshape(4, 4),
            "campos": torch.tensor(list(fr.campos), dtype=torch.float32), "tanfovx": float(fr.tanfovx), "tanfovy": float(fr.tanfovy)}


def oracle_activated(oracle, gm, delta_row, min_kernel_size=synthetic.KERNEL_3D):
    """GaussianModel.get_*_with_delta on the CPU oracle -> the attribute dict oracle_render() takes."""
    n = lambda t: None if t is None else t.detach().cpu().numpy()
    oa = oracle.gaussian_activate(n(gm._xyz), n(gm.get_features), n(gm._scaling), n(gm._rotation), n(gm._opacity), n(delta_row),
                                  aabb=[-0.5, -0.5, -0.5, 1, 1, 1], scale_bias=float(gm.scale_bias), opacity_bias=float(gm.opacity_bias),
                                  min_kernel_size=min_kernel_size, scaling_activation=1)
    attrs = {k: torch.from_numpy(oa[k]) for k in ("means3D", "scales", "rotations", "shs")}
    attrs["opacities"] = torch.from_numpy(oa["opacities"])
    return attrs


# Tolerance of the rasteriser parity tests (BASELINE.json north_star: "rendered RGBA frames must
# match ... within 1e-3 max-abs per pixel").
RAST_ATOL = 1e-3


FLIP_ATOL = 5e-3       # one splat at the alpha threshold skipped or added: <= (1/255) * colour (<= 1) * T (<= 1) = 3.9e-3, + margin


def compare_images(hip: np.ndarray, ref: np.ndarray, flags: np.ndarray, atol=RAST_ATOL, max_flag_frac=0.03,
                   flagged_atol=RAST_ATOL, max_flips=None, flip_atol=FLIP_ATOL):
    """max-abs <= atol (north_star: 1e-3) on EVERY pixel, flagged or not, with one exemption: a flagged pixel (oracle `flags` != 0: one of its
    blend decisions sits within float noise of a threshold -- alpha ~ 1/255, T ~ 1e-4, power ~ 0; the oracle uses libm expf, the device
    v_exp_f32) may have taken the other side of that decision, i.e. differ by one skipped / added splat (<= FLIP_ATOL).  Such flips are
    counted, printed and bounded: at most max(2, 1e-4 of the pixels) of them.  (Round 3 let every flagged pixel through at 2e-2; the measured
    flagged error is ~2e-7, so the exemption now covers real flips only.)
    Returns (max_err_unflagged, max_err_flagged, flagged_fraction)."""
    err = np.abs(hip - ref)
    if err.ndim == 3:
        err = err.max(axis=0)
    clean = flags == 0
    e_clean = float(err[clean].max()) if clean.any() else 0.0
    e_flag = float(err[~clean].max()) if (~clean).any() else 0.0
    frac = float((~clean).mean())
    assert e_clean <= atol, f"unflagged pixels differ by {e_clean} > {atol}"
    assert frac <= max_flag_frac, f"{frac:.4f} of the pixels are threshold-flagged"
    flips = (~clean) & (err > flagged_atol)
    n_flips = int(flips.sum())
    if n_flips:
        limit = max(2, int(1e-4 * err.size)) if max_flips is None else max_flips
        worst = float(err[flips].max())
        print(f"compare_images: {n_flips} of {int((~clean).sum())} flagged pixels took the other side of a threshold decision (max {worst:.2e}); "
              f"every other pixel <= {atol}")
        assert n_flips <= limit, f"{n_flips} flagged pixels differ by more than {flagged_atol} (limit {limit})"
        assert worst <= flip_atol, f"a flagged pixel differs by {worst} > one threshold splat ({flip_atol})"
    return e_clean, e_flag, frac
