"""Real spherical-harmonics basis, degrees 0-3, with the reference's conventions
(renderers/sh_utils.py:26-112: same coefficient ordering and signs; RGB2SH/SH2RGB :114-117).
Host-side helper for the `convert_SHs_python` branch and for building synthetic inputs; the
rasteriser evaluates SH in its preprocess kernel (csrc/rast.hip sh_to_rgb)."""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """(..., 3) unit directions -> (..., (deg+1)^2) basis values."""
    assert 0 <= deg <= 3
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    b = [torch.full_like(x, C0)]
    if deg > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz = x * x, y * y, z * z
        b += [C2[0] * x * y, C2[1] * y * z, C2[2] * (2.0 * zz - xx - yy), C2[3] * x * z, C2[4] * (xx - yy)]
    if deg > 2:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * x * y * z, C3[2] * y * (4 * zz - xx - yy),
              C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
              C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=-1)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh (..., C, >= (deg+1)^2), dirs (..., 3) -> (..., C)."""
    n = (deg + 1) ** 2
    assert sh.shape[-1] >= n
    return (sh[..., :n] * sh_basis(deg, dirs)[..., None, :]).sum(-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
