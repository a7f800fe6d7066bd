"""SparseVAE -- the static-VAE framework object the inference / encoding scripts hold
(model/sparse_voxel_diffusion/sparse_vae.py:60-485): owns the backbone, converts its per-voxel output rows to
GaussianModel objects (`to_representation`), owns the renderers (`static_vae.renderers["MipGS"]`,
inference_dpm_latent.py:161, utils/inference_utils.py:52-65) and exposes encode / decode / encode_decode[_no_render] /
render_batch.  The loss / regularisation / snapshot members are training code and are not built."""
import copy
from typing import *

import torch

from ...attrdict import edict
from ...renderers import GaussianRenderer
from ...representations.gaussian.voxel_rows import gaussian_row_layout, rows_to_gaussian

__all__ = ["SparseVAE", "hammersley_sequence"]

_DEFAULT_GAUSSIAN_LR_CONFIG = {"_xyz": 1.0, "_features_dc": 0.0025, "_opacity": 0.05, "_scaling": 0.005, "_rotation": 0.001}
_DEFAULT_GS_CFG = {"lr": _DEFAULT_GAUSSIAN_LR_CONFIG, "perturb_offset": False, "reg_mode": "invoxel", "voxel_size": 1.1,
                   "num_gaussians": 8, "scaling_bias": 0.01, "opacity_bias": 0.1, "scaling_activation": "exp"}
_DEFAULT_MIPGS_CFG = dict(_DEFAULT_GS_CFG, **{"2d_filter_kernel_size": 0.1, "3d_filter_kernel_size": 0.0})
_DEFAULT_CONFIG = {"GS": _DEFAULT_GS_CFG, "MipGS": _DEFAULT_MIPGS_CFG}

_PRIMES = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53]


def _radical_inverse(base: int, n: int) -> float:
    val, inv, scale = 0.0, 1.0 / base, 1.0 / base
    while n > 0:
        val += (n % base) * scale
        n //= base
        scale *= inv
    return val


def hammersley_sequence(dim: int, n: int, num_samples: int) -> List[float]:
    """model/sparse_voxel_diffusion/utils.py:62-78."""
    return [n / num_samples] + [_radical_inverse(_PRIMES[d], n) for d in range(dim - 1)]


def _unwrap(model):
    return model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model


class SparseVAE:
    def __init__(self, backbones, resolution=64, representation_config={}, loss_type="l1", lambda_ssim=0.2, lambda_lpips=0.2,
                 lamda_kl=1e-6, regularizations={}, mem_ratio=1.0):
        assert "vae" in backbones.keys(), "A VAE backbone must be provided."
        self.backbones = backbones
        self.resolution = resolution
        self.loss_type = loss_type
        self.lambda_ssim = lambda_ssim
        self.lambda_lpips = lambda_lpips
        self.lamda_kl = lamda_kl
        self.mem_ratio = mem_ratio
        self.regularizations = regularizations
        self.rep_config = {}
        for k, v in representation_config.items():
            if k not in _DEFAULT_CONFIG:
                raise ValueError(f"Invalid representation type: {k}")
            self.rep_config[k] = copy.deepcopy(_DEFAULT_CONFIG[k])
            self.rep_config[k].update(v)
        self._init_renderer()
        self._calc_layout(self.rep_config)
        for k in ("GS", "MipGS"):
            if k in self.rep_config and self.rep_config[k]["perturb_offset"]:
                _unwrap(self.backbones["vae"]).register_buffer(
                    f"{k}_perturbation", self._build_perturbation(self.rep_config[k]["num_gaussians"], self.rep_config[k]["reg_mode"]))

    def get_phases(self, step):
        return ["vae"]

    def _build_perturbation(self, num_gaussians, reg_mode):
        offsets = torch.tensor([hammersley_sequence(3, i, num_gaussians) for i in range(num_gaussians)]).float() - 0.5
        if reg_mode == "soft_invoxel":
            offsets = offsets / 0.5 / self.rep_config["MipGS"]["voxel_size"]      # (:110; the MipGS voxel size whatever k is)
        return torch.atanh(offsets).to(_unwrap(self.backbones["vae"]).device)

    def _calc_layout(self, rep_config):
        self.layouts, start = {}, 0
        for kind, cfg in rep_config.items():                 # channel ranges run on across the representation kinds (:202-227)
            self.layouts[kind] = gaussian_row_layout(cfg["num_gaussians"], start)
            start = self.layouts[kind]["_opacity"]["range"][1]
        self.layouts = edict(self.layouts)

    def get_renderer(self, type, rendering_options):
        renderer = GaussianRenderer(rendering_options)
        if type == "MipGS":
            renderer.pipe.use_mip_gaussian = True
            renderer.pipe.kernel_size = self.rep_config["MipGS"]["2d_filter_kernel_size"]
        elif type != "GS":
            raise ValueError(f"Invalid representation type: {type}")
        return renderer

    def _init_renderer(self):
        rendering_options = {"near": 0.8, "far": 1.6, "bg_color": (1.0, 1.0, 1.0)}
        self.renderers = edict({k: self.get_renderer(k, rendering_options) for k in self.rep_config.keys()})

    def to_representation(self, x):
        """(N x * x C) sparse output rows -> {kind: [GaussianModel per sample]}  (:114-182)."""
        ret = {k: [] for k in self.rep_config.keys()}
        vae = _unwrap(self.backbones["vae"])
        for sl in x.layout:
            for kind, cfg in self.rep_config.items():
                # offsets: tanh / resolution inside the voxel ("invoxel"), or half a `voxel_size` ("soft_invoxel"; the GS kind
                # hard-codes 1.25 there, :139)
                scale = {"invoxel": 1.0, "soft_invoxel": 0.5 * (cfg["voxel_size"] if kind == "MipGS" else 1.25)}.get(cfg["reg_mode"])
                if scale is None:
                    raise ValueError(f"unknown reg_mode {cfg['reg_mode']}")
                kw = dict(mininum_kernel_size=cfg.get("3d_filter_kernel_size", 0.0) if kind == "MipGS" else 0.0,
                          scaling_bias=cfg["scaling_bias"], opacity_bias=cfg["opacity_bias"], scaling_activation=cfg["scaling_activation"])
                ret[kind].append(rows_to_gaussian(x.feats[sl], x.coords[sl][:, 1:], self.resolution, self.layouts[kind], cfg["lr"], scale,
                                                  getattr(vae, f"{kind}_perturbation") if cfg["perturb_offset"] else None, kw))
        return ret

    def render_batch(self, reps, extrinsics: torch.Tensor, intrinsics: torch.Tensor):
        ret = {k: None for k in self.rep_config.keys()}
        for k, v in reps.items():
            for i, representation in enumerate(v):
                pack = self.renderers[k].render(representation, extrinsics[i], intrinsics[i])
                if ret[k] is None:
                    ret[k] = {kk: [] for kk in list(pack.keys()) + ["bg_color"]}
                for kk, vv in pack.items():
                    ret[k][kk].append(vv)
                ret[k]["bg_color"].append(self.renderers[k].bg_color)
            for kk, vv in ret[k].items():
                ret[k][kk] = torch.stack(vv, dim=0)
        return ret

    def encode_decode(self, feats, image, extrinsics, intrinsics, return_aux=False, **kwargs):
        x, mean, logvar = self.backbones["vae"](feats, mem_ratio=self.mem_ratio)
        reps = self.to_representation(x)
        for v in self.renderers.values():
            v.rendering_options.resolution = image.shape[-1]
        render_results = self.render_batch(reps, extrinsics, intrinsics)
        rec_image = torch.cat([v["rgb"] for v in render_results.values()])
        gt_image = torch.cat([image for _ in render_results.values()])
        if return_aux:
            return reps, {"x": x, "rec_image": rec_image, "gt_image": gt_image, "mean": mean, "logvar": logvar}
        return reps

    def encode_decode_no_render(self, feats, return_aux=False, **kwargs):
        x, mean, logvar = self.backbones["vae"](feats, mem_ratio=self.mem_ratio)
        reps = self.to_representation(x)
        if return_aux:
            return reps, {"x": x, "mean": mean, "logvar": logvar}
        return reps

    def encode(self, feats, **kwargs):
        return self.backbones["vae"].encode(feats, **kwargs)

    def decode(self, latent):
        return self.to_representation(self.backbones["vae"].decode(latent))

    def training_losses(self, *a, **k):
        raise NotImplementedError("static-VAE training losses (l1 / ssim / lpips, regularisers) are outside the hot path")
