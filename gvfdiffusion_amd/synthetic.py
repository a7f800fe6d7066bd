"""Synthetic inputs of the shapes BASELINE.json names: random Gaussians,
orbit cameras with the reference's conventions, per-frame deltas.  There are no datasets or
checkpoints in this environment; every bench and test input comes from here (seeded, CPU
generator, then moved to the device)."""
import math

import torch

from .renderers.sh_utils import RGB2SH
from .representations.gaussian import Gaussian, GaussianModel

FOV_X_DEG = 49.1  # dataset/dataset_latent_inference.py:182
NEAR, FAR = 0.8, 1.6  # model/sparse_voxel_diffusion/sparse_vae.py:197-199
BG = (1.0, 1.0, 1.0)
KERNEL_2D = 0.1  # configs/diffusion.yml:81 (2d_filter_kernel_size)
KERNEL_3D = 0.0009
SCALING_BIAS = 0.004
OPACITY_BIAS = 0.1


def orbit_w2c(azimuth_deg: float, elevation_deg: float = 0.0, radius: float = 2.0) -> torch.Tensor:
    """World-to-camera (4,4) of a camera orbiting the origin in a z-up world, COLMAP axes
    (x right, y down, z forward) -- the convention the reference's cameras end in
    (utils/inference_utils.py:245-254: orbit pose -> y-up to z-up -> flip y,z -> invert)."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    eye = torch.tensor([radius * math.cos(el) * math.sin(az), -radius * math.cos(el) * math.cos(az),
                        radius * math.sin(el)], dtype=torch.float64)
    fwd = -eye / eye.norm()
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    c2w = torch.eye(4, dtype=torch.float64)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, eye
    return torch.linalg.inv(c2w).float()


def intrinsics(fov_x_deg: float = FOV_X_DEG) -> torch.Tensor:
    f = 0.5 / math.tan(math.radians(fov_x_deg) / 2)
    return torch.tensor([[f, 0, 0.5], [0, f, 0.5], [0, 0, 1]], dtype=torch.float32)


def camera_block(azi=0.0, elev=0.0, radius=2.0, fov=FOV_X_DEG, near=NEAR, far=FAR):
    """One orbit camera in every form the rasteriser's callers use (renderers/gaussian_render.py:285-321): extrinsics (w2c), normalised
    intrinsics, and the derived GaussianRasterizationSettings fields (viewmatrix = V^T, projmatrix = (P V)^T, campos, tan fov)."""
    from .renderers.gaussian_render import intrinsics_to_projection
    view = orbit_w2c(azi, elev, radius)
    K = intrinsics(fov)
    persp = intrinsics_to_projection(K, near, far)
    tan = math.tan(float(2 * torch.atan(0.5 / K[0, 0])) * 0.5)
    return dict(extrinsics=view, intrinsics=K, viewmatrix=view.T.contiguous(),
                projmatrix=(persp @ view).T.contiguous(), campos=torch.inverse(view)[:3, 3].contiguous(),
                tanfovx=tan, tanfovy=tan)


def random_gaussians(P: int, sh_degree: int = 2, seed: int = 0, scale_lo: float = 0.003, scale_hi: float = 0.02):
    """Activated attributes (dict of CPU fp32 tensors): means3D, scales, rotations, opacities (P,1), shs (P,M,3)."""
    g = torch.Generator().manual_seed(seed)
    M = (sh_degree + 1) ** 2
    xyz = torch.rand((P, 3), generator=g) - 0.5
    scales = torch.exp(torch.rand((P, 3), generator=g) * (math.log(scale_hi) - math.log(scale_lo)) + math.log(scale_lo))
    rot = torch.nn.functional.normalize(torch.randn((P, 4), generator=g), dim=1)
    opac = torch.sigmoid(torch.randn((P, 1), generator=g) * 1.5)
    shs = torch.randn((P, M, 3), generator=g) * 0.1
    shs[:, 0] = RGB2SH(torch.rand((P, 3), generator=g))
    return dict(means3D=xyz, scales=scales, rotations=rot, opacities=opac, shs=shs)


def gaussian_model_from(attrs, sh_degree: int, device, scaling_activation="softplus") -> GaussianModel:
    """GaussianModel whose activated accessors reproduce `attrs` (inverse activations applied)."""
    # the TRELLIS twin: all SH bands active from the start (the mirror of representations/gaussian starts at degree 0)
    gm = Gaussian(sh_degree=sh_degree, aabb=[-0.5, -0.5, -0.5, 1.0, 1.0, 1.0], mininum_kernel_size=KERNEL_3D,
                       scaling_bias=SCALING_BIAS, opacity_bias=OPACITY_BIAS, scaling_activation=scaling_activation,
                       device=device)
    d = {k: v.to(device) for k, v in attrs.items()}
    gm.from_xyz(d["means3D"])
    gm.from_scaling(d["scales"])
    gm.from_rotation(d["rotations"])
    gm.from_features(d["shs"].contiguous())
    gm.from_opacity(d["opacities"].clamp(1e-4, 1 - 1e-4))
    return gm


def random_deltas(T: int, P: int, seed: int = 1, std: float = 0.01) -> torch.Tensor:
    """(T,P,14) per-frame deltas ~ N(0,std) on all channels [xyz3|scale3|rot4|rgb3|op1]."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn((T, P, 14), generator=g) * std


# ---- DiT (configs/diffusion.yml) -----------------------------------------------------------------
def dit_state_dict(manifest: dict, seed: int = 0) -> dict:
    """Deterministic full-size DiT weights from a {name: shape} manifest (tests/golden/dit_manifest.json):
    no checkpoint exists in this environment.  Matrices ~ N(0, 1/fan_in) so activations stay O(1) through 12
    blocks, biases ~ N(0, 0.02^2), norm gains 1 + 0.1 N(0,1).  One generator per tensor, seeded by its
    index in sorted-name order, so the reference and this package build bit-identical tensors."""
    sd = {}
    for idx, name in enumerate(sorted(manifest)):
        shape = tuple(manifest[name])
        g = torch.Generator().manual_seed(seed * 100003 + idx)
        if name.endswith("gamma") or (name.endswith("weight") and len(shape) == 1):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 2:
            w = torch.randn(shape, generator=g) / math.sqrt(shape[1])
            if "adaLN_modulation" in name:
                w = w * 0.2
        else:
            w = 0.02 * torch.randn(shape, generator=g)
        sd[name] = w
    return sd


def dit_state_dict_trained_like(manifest: dict, seed: int = 0, gamma_lo: float = 0.5, gamma_hi: float = 2.0, cross_gain: float = 1.3,
                                outlier_gain: float = 1.6) -> dict:
    """dit_state_dict() pushed towards the score statistics of a TRAINED denoiser (none exists in this environment): the seed-generated
    weights give every attention near-uniform scores (std ~ 1), which is the friendliest case for the tiled attention's max-free softmax.
      * every QK-RMSNorm gain (model/attention/modules.py:8-15, `*.q_rms_norm.gamma` / `*.k_rms_norm.gamma`) ~ U[gamma_lo, gamma_hi]: the
        self attentions' log2-domain scores reach 5.66 * gamma_hi^2 * 1.44, outside the `scores_bounded` promise (fp16 takes its shift);
      * the cross attentions run WITHOUT RMSNorm at qk_rms_norm_cross=False (modules.py:134-143): their `to_q` and the k half of `to_kv`
        are scaled by cross_gain each, i.e. scores by cross_gain^2;
      * heavy tails: per cross attention a rank-one term  outlier_gain * u a^T  is added to `to_q` and  outlier_gain * u b^T  to the k rows of
        `to_kv` (u: one unit direction per head inside that head's 32 dims, a / b: unit directions of the normalised stream / the context):
        the score gains  outlier_gain^2 * (a . x)(b . c),  a PRODUCT of two near-normal variables -- most pairs move little, a few (query,
        key) pairs land tens of octaves out, in any key tile (the values are untouched, so the stream stays O(1))."""
    sd = dit_state_dict(manifest, seed)
    for idx, name in enumerate(sorted(sd)):
        g = torch.Generator().manual_seed(seed * 100003 + 50021 + idx)
        if name.endswith("rms_norm.gamma"):
            sd[name] = gamma_lo + (gamma_hi - gamma_lo) * torch.rand(sd[name].shape, generator=g)
        elif "cross_attn.to_q.weight" in name or "cross_attn.to_kv.weight" in name:
            w = sd[name]
            C = w.shape[0] if "to_q" in name else w.shape[0] // 2
            H = C // 32
            gu = torch.Generator().manual_seed(seed * 100003 + 70001 + sorted(sd).index(name.replace("to_kv", "to_q")))     # the SAME u for q and k
            u = torch.nn.functional.normalize(torch.randn((H, 32), generator=gu), dim=-1).reshape(C)
            d_in = torch.nn.functional.normalize(torch.randn((w.shape[1],), generator=g), dim=0)
            # x, c have unit-variance components, so a . x and b . c are ~ N(0, 1): extra score = outlier_gain^2 * N * N' (natural-log units)
            k_rows = w[:C] * cross_gain + (outlier_gain * 32.0 ** 0.25) * torch.outer(u, d_in)
            sd[name] = k_rows if "to_q" in name else torch.cat([k_rows, w[C:]])          # rows [0, C) = k, [C, 2C) = v   (kv.reshape(B, L, 2, H, d))
    return sd


def dit_inputs_hostile(B: int = 1, T: int = 24, seed: int = 1, token_gain: float = 2.5, **kw) -> dict:
    """dit_inputs() with a few high-norm context tokens, the artefact tokens a DINOv2 backbone is known for: three image tokens per frame
    and three static tokens carry token_gain x the typical norm (their keys AND values).  Positions: the first, a middle and the last
    64-key tile of each context."""
    inp = dit_inputs(B=B, T=T, seed=seed, **kw)
    Li, Ls = inp["cond_images"].shape[2], inp["static_latent"].shape[1]
    inp["cond_images"][:, :, [5, Li // 2 + 15, Li - 3]] *= token_gain
    inp["static_latent"][:, [17, Ls // 2 + 174, Ls - 96]] *= token_gain
    return inp


def dit_inputs(B: int = 1, T: int = 24, N: int = 512, L_img: int = 1370, L_static: int = 4096, C: int = 16,
               img_channels: int = 1024, static_channels: int = 14, seed: int = 1) -> dict:
    """BASELINE configs[2] inputs: x ~ N(0,1) (B,T,N,16), DINOv2-shaped cond_images (B,T,1370,1024),
    static_latent (B,4096,14), deformation_position_xyz ~ U(-.5,.5) (B,N,3), t (B,)."""
    g = torch.Generator().manual_seed(seed)
    return dict(x=torch.randn((B, T, N, C), generator=g),
                t=torch.full((B,), 498.996),
                cond_images=torch.randn((B, T, L_img, img_channels), generator=g),
                static_latent=torch.randn((B, L_static, static_channels), generator=g),
                deformation_position_xyz=torch.rand((B, N, 3), generator=g) - 0.5)
