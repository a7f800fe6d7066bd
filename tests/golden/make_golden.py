"""Generates the committed golden fixtures by importing the REFERENCE (read-only, /root/reference).

Runs only in the build container (the reference does not travel to the GPU box); the .npz files it
writes under tests/golden/ are data: inputs + the reference's outputs.  Usage:
    python tests/golden/make_golden.py [raster] [vox2seq] [dit] [sampler] [sparse]   (default: all)

Stubs (throw-away, module level, nothing shipped): the reference hard-imports packages that are not
in this image -- spconv (sparse/basic.py:6), flash_attn (model/sparse_attention/full_attn.py:8-9),
easydict / utils3d / plyfile (renderers, representations) -- and calls .cuda() in
GaussianModel.setup_functions (representations/gaussian/gaussian_model.py:38-41); those are
neutralised here so that the reference's own Python code runs on CPU fp32.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    os.environ["ATTN_BACKEND"] = "naive"           # model/attention/__init__.py:12-20
    os.environ["SPARSE_ATTN_BACKEND"] = "flash_attn"
    sp = _stub("spconv"); spp = _stub("spconv.pytorch", SparseConvTensor=type("SparseConvTensor", (), {}))
    sp.pytorch = spp
    _stub("flash_attn")
    _stub("utils3d")
    _stub("plyfile", PlyData=object, PlyElement=object)

    class EasyDict(dict):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.__dict__ = self
    _stub("easydict", EasyDict=EasyDict)
    torch.Tensor.cuda = lambda self, *a, **k: self  # GaussianModel.setup_functions on CPU
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load_by_path(name, path, search=None):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=search)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------
def gen_raster():
    g = torch.Generator().manual_seed(1234)
    out = {}
    sh_utils = load_by_path("ref_sh_utils", f"{REF}/renderers/sh_utils.py")
    n = 64
    dirs = torch.nn.functional.normalize(torch.randn((n, 3), generator=g), dim=1)
    coeffs = torch.randn((n, 3, 16), generator=g)
    out["sh_dirs"], out["sh_coeffs"] = dirs.numpy(), coeffs.numpy()
    for deg in range(4):
        out[f"sh_out_deg{deg}"] = sh_utils.eval_sh(deg, coeffs, dirs).numpy()

    from utils.script_util import build_rotation  # CPU twin of general_utils.build_rotation
    quats = torch.randn((32, 4), generator=g)
    out["quats"], out["rotmats"] = quats.numpy(), build_rotation(quats).numpy()

    from renderers.gaussian_render import intrinsics_to_projection
    K = torch.tensor([[1.0946, 0, 0.5], [0, 1.0946, 0.5], [0, 0, 1]])
    out["intrinsics"], out["projection"] = K.numpy(), intrinsics_to_projection(K, 0.8, 1.6).numpy()

    # GaussianModel activations with deltas (gaussian_model.py:84-114), the real class on CPU
    from representations.gaussian import GaussianModel
    P = 96
    gm = GaussianModel(sh_degree=0, aabb=[-0.5, -0.5, -0.5, 1.0, 1.0, 1.0], mininum_kernel_size=0.0009,
                       scaling_bias=0.004, opacity_bias=0.1, scaling_activation="softplus", device="cpu")
    gm._xyz = torch.rand((P, 3), generator=g)
    gm._features_dc = torch.randn((P, 1, 3), generator=g)
    gm._scaling = torch.randn((P, 3), generator=g) * 2
    gm._scaling[0, 0] = 30.0   # softplus threshold branch
    gm._rotation = torch.randn((P, 4), generator=g)
    gm._opacity = torch.randn((P, 1), generator=g) * 3
    delta = torch.randn((P, 14), generator=g) * 0.1
    out["act_xyz"], out["act_feat"], out["act_scaling"] = gm._xyz.numpy(), gm._features_dc.numpy(), gm._scaling.numpy()
    out["act_rot"], out["act_opacity"], out["act_delta"] = gm._rotation.numpy(), gm._opacity.numpy(), delta.numpy()
    out["act_scale_bias"] = np.float32(gm.scale_bias.item())
    out["act_opacity_bias"] = np.float32(gm.opacity_bias.item())
    out["act_out_means3D"] = gm.get_xyz_with_delta(delta[:, :3]).numpy()
    out["act_out_scales"] = gm.get_scaling_with_delta(delta[:, 3:6]).numpy()
    out["act_out_rotations"] = gm.get_rotation_with_delta(delta[:, 6:10]).numpy()
    out["act_out_shs"] = gm.get_features_with_delta(delta[:, 10:13].unsqueeze(1)).numpy()
    out["act_out_opacities"] = gm.get_opacity_with_delta(delta[:, 13:]).numpy()
    np.savez_compressed(os.path.join(OUT, "raster_mirrors.npz"), **out)
    print("raster_mirrors.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def gen_vox2seq():
    base = f"{REF}/model/sparse_voxel_diffusion/vox2seq/vox2seq/pytorch"
    v = load_by_path("ref_vox2seq_pt", f"{base}/__init__.py", [base])
    g = torch.Generator().manual_seed(7)
    coords = torch.randint(0, 1024, (4096, 3), generator=g, dtype=torch.int32)
    special = torch.tensor([[1, 0, 0], [0, 1, 0], [0, 0, 1], [3, 5, 7], [1023, 1023, 1023], [63, 0, 12], [0, 0, 0]],
                           dtype=torch.int32)
    coords = torch.cat([special, coords])
    out = {"coords": coords.numpy()}
    for mode in ("z_order", "hilbert"):
        code = v.encode(coords, mode=mode)
        out[f"{mode}_code"] = code.numpy().astype(np.int64)
        out[f"{mode}_decode_of_0_63"] = v.decode(torch.arange(64), mode=mode).numpy()
        assert torch.equal(v.decode(code, mode=mode).int(), coords)
    np.savez_compressed(os.path.join(OUT, "vox2seq_golden.npz"), **out)
    print("vox2seq_golden.npz", {k: v.shape for k, v in out.items()})


SECTIONS = {"raster": gen_raster, "vox2seq": gen_vox2seq}

if __name__ == "__main__":
    install_stubs()
    todo = sys.argv[1:] or list(SECTIONS)
    for name in todo:
        SECTIONS[name]()
