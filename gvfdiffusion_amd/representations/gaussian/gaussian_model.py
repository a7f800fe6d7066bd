"""Gaussian container with the reference's accessor surface.

Mirrors representations/gaussian/gaussian_model.py:15-128 (and its TRELLIS twin
trellis/representations/gaussian/gaussian_model.py:8-113, the object actually passed at inference):
aabb-normalised xyz, exp/softplus scale with bias and 3D filter, sigmoid opacity with bias,
quaternion + (1,0,0,0) bias, and the *_with_delta accessors the renderer uses.  Unlike the
reference it is device-agnostic (no hard-coded .cuda()), and `activation_struct()` hands the same
constants to the fused HIP path (csrc/rast.hip, GvfGaussianActivation).  PLY IO is out of scope
(utils3d/plyfile are IO, not on the hot path).
"""
import math

import torch
import torch.nn.functional as F


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """Unit-normalise (r,x,y,z) and return (N,3,3) rotation matrices (general_utils.py:78-99)."""
    q = r / torch.linalg.vector_norm(r, dim=1, keepdim=True)
    w, x, y, z = q.unbind(dim=1)
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def build_scaling_rotation(s: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """L = R diag(s) (general_utils.py:101-110)."""
    return build_rotation(r) * s[:, None, :]


def strip_symmetric(sym: torch.Tensor) -> torch.Tensor:
    """Upper triangle (xx,xy,xz,yy,yz,zz) of (N,3,3) (general_utils.py:64-76)."""
    return torch.stack([sym[:, 0, 0], sym[:, 0, 1], sym[:, 0, 2], sym[:, 1, 1], sym[:, 1, 2], sym[:, 2, 2]], dim=1)


def _inv_softplus(x: float) -> float:
    return x + math.log(-math.expm1(-x))


class GaussianModel:
    _TRELLIS_TWIN = False

    def __init__(self, sh_degree: int = 0, aabb=(-0.5, -0.5, -0.5, 1.0, 1.0, 1.0), mininum_kernel_size: float = 0.0,
                 scaling_bias: float = 0.01, opacity_bias: float = 0.1, scaling_activation: str = "exp",
                 device="cuda"):
        if scaling_activation not in ("exp", "softplus"):
            raise ValueError(f"unknown scaling activation {scaling_activation}")
        self.init_params = dict(aabb=list(aabb), sh_degree=sh_degree, mininum_kernel_size=mininum_kernel_size,
                                scaling_bias=scaling_bias, opacity_bias=opacity_bias,
                                scaling_activation=scaling_activation)
        self.sh_degree = sh_degree
        self.max_sh_degree = sh_degree
        # representations/gaussian/gaussian_model.py:54 starts at 0 and relies on oneupSHdegree(); the TRELLIS twin
        # (trellis/representations/gaussian/gaussian_model.py:29, class `Gaussian` below) starts at sh_degree.  Every use in the
        # reference has sh_degree 0, where the two agree.
        self.active_sh_degree = sh_degree if self._TRELLIS_TWIN else 0
        self.mininum_kernel_size = mininum_kernel_size
        self.scaling_bias = scaling_bias
        self.scaling_activation_type = scaling_activation
        self.device = torch.device(device)
        self.aabb = torch.tensor(list(aabb), dtype=torch.float32, device=self.device)
        if scaling_activation == "exp":
            self.scaling_activation = torch.exp
            self.inverse_scaling_activation = torch.log
            sb = math.log(scaling_bias)
        else:
            self.scaling_activation = F.softplus
            self.inverse_scaling_activation = lambda x: x + torch.log(-torch.expm1(-x))
            sb = _inv_softplus(scaling_bias)
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = lambda x: torch.log(x / (1 - x))
        self.rotation_activation = F.normalize
        # biases are computed in fp32 exactly as the reference does (tensor ops on a fp32 scalar)
        self.scale_bias = self.inverse_scaling_activation(torch.tensor(scaling_bias, dtype=torch.float32)).to(self.device)
        self.rots_bias = torch.tensor([1.0, 0.0, 0.0, 0.0], device=self.device)
        self.opacity_bias = self.inverse_opacity_activation(torch.tensor(opacity_bias, dtype=torch.float32)).to(self.device)
        del sb
        self._xyz = None
        self._features_dc = None
        self._features_rest = None
        self._scaling = None
        self._rotation = None
        self._opacity = None

    # ---- activated accessors ---------------------------------------------------------------
    def _scale_from(self, raw):
        s = self.scaling_activation(raw)
        return torch.sqrt(torch.square(s) + self.mininum_kernel_size ** 2)

    @property
    def get_scaling(self):
        return self._scale_from(self._scaling + self.scale_bias)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation + self.rots_bias[None, :])

    @property
    def get_xyz(self):
        return self._xyz * self.aabb[None, 3:] + self.aabb[None, :3]

    @property
    def get_features(self):
        # representations/gaussian/gaussian_model.py:117-121 returns the DC term only; the TRELLIS twin (:87-88) appends the rest on dim 2
        if self._TRELLIS_TWIN and self._features_rest is not None:
            return torch.cat((self._features_dc, self._features_rest), dim=2)
        return self._features_dc

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity + self.opacity_bias)

    def get_xyz_with_delta(self, delta, detach=False):
        base = self.get_xyz.detach() if detach else self.get_xyz
        return base + delta

    def get_scaling_with_delta(self, delta, detach=False):
        raw = self._scaling.detach() if detach else self._scaling
        return self._scale_from(raw + self.scale_bias + delta)

    def get_rotation_with_delta(self, delta, detach=False):
        raw = self._rotation.detach() if detach else self._rotation
        return self.rotation_activation(raw + self.rots_bias[None, :] + delta)

    def get_features_with_delta(self, delta, detach=False):
        raw = self._features_dc.detach() if detach else self._features_dc
        return raw + delta

    def get_opacity_with_delta(self, delta, detach=False):
        raw = self._opacity.detach() if detach else self._opacity
        return self.opacity_activation(raw + self.opacity_bias + delta)

    def get_covariance(self, scaling_modifier=1):
        L = build_scaling_rotation(scaling_modifier * self.get_scaling, self._rotation + self.rots_bias[None, :])
        return strip_symmetric(L @ L.transpose(1, 2))

    # ---- setters from activated values -----------------------------------------------------
    def from_scaling(self, scales):
        scales = torch.sqrt(torch.square(scales) - self.mininum_kernel_size ** 2)
        self._scaling = self.inverse_scaling_activation(scales) - self.scale_bias

    def from_rotation(self, rots):
        self._rotation = rots - self.rots_bias[None, :]

    def from_xyz(self, xyz):
        self._xyz = (xyz - self.aabb[None, :3]) / self.aabb[None, 3:]

    def from_features(self, features):
        self._features_dc = features

    def from_opacity(self, opacities):
        self._opacity = self.inverse_opacity_activation(opacities) - self.opacity_bias

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- fused HIP path --------------------------------------------------------------------
    def activation_struct(self):
        from ... import _lib
        a = _lib.GvfGaussianActivation()
        ab = self.aabb.detach().cpu().tolist()
        for k in range(6):
            a.aabb[k] = ab[k]
        a.scale_bias = float(self.scale_bias)
        a.opacity_bias = float(self.opacity_bias)
        a.min_kernel_size = float(self.mininum_kernel_size)
        a.scaling_activation = 0 if self.scaling_activation_type == "exp" else 1
        return a


class Gaussian(GaussianModel):
    """trellis/representations/gaussian/gaussian_model.py: the same class with active_sh_degree = sh_degree from the start and
    get_features = DC | rest (the structured-latent decoder's output type)."""
    _TRELLIS_TWIN = True
