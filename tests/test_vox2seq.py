"""Z-order / Hilbert voxel serialisation: the C oracle against the reference's known answers and golden
codes (tests/golden/vox2seq_golden.npz, produced by the reference's vox2seq/pytorch fallback -- the thing
vox2seq/test.py asserts its CUDA extension equals), and the HIP kernels against both."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vox2seq_golden.npz"))


def test_oracle_known_answers_and_golden(oracle_lib):
    c = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [3, 5, 7], [1023, 1023, 1023], [63, 0, 12]])
    assert oracle_lib.vox2seq_encode(c, "z_order").tolist() == [4, 2, 1, 239, 1073741823, 150372]
    assert oracle_lib.vox2seq_encode(c, "hilbert").tolist() == [7, 3, 1, 391, 766958445, 121005]
    assert oracle_lib.vox2seq_decode(np.arange(9), "hilbert").tolist() == \
        [[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0], [1, 1, 0], [1, 1, 1], [1, 0, 1], [1, 0, 0], [2, 0, 0]]
    for mode in ("z_order", "hilbert"):
        assert np.array_equal(oracle_lib.vox2seq_encode(G["coords"], mode), G[f"{mode}_code"].astype(np.int32))
        assert np.array_equal(oracle_lib.vox2seq_decode(np.arange(64), mode), G[f"{mode}_decode_of_0_63"].astype(np.int32))
        assert np.array_equal(oracle_lib.vox2seq_decode(G[f"{mode}_code"], mode), G["coords"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["z_order", "hilbert"])
def test_hip_kernels_bit_exact(cuda, oracle_lib, mode):
    from gvfdiffusion_amd.sparse import vox2seq
    coords = torch.from_numpy(G["coords"]).to(cuda)
    code = vox2seq.encode(coords, mode=mode)
    assert code.dtype == torch.int32 and np.array_equal(code.cpu().numpy(), G[f"{mode}_code"].astype(np.int32))
    assert torch.equal(vox2seq.decode(code, mode=mode), coords)
    # the reference's own test shape: the full 256^3 grid, round trip + agreement with the CPU oracle (vox2seq/test.py)
    r = torch.arange(256, dtype=torch.int32)
    grid = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), dim=-1).reshape(-1, 3)
    gcode = vox2seq.encode(grid.to(cuda), mode=mode)
    assert np.array_equal(gcode.cpu().numpy(), oracle_lib.vox2seq_encode(grid.numpy(), mode))
    assert gcode.unique().numel() == 256 ** 3 and int(gcode.max()) == 256 ** 3 - 1      # a bijection onto [0, 2^24)
    assert torch.equal(vox2seq.decode(gcode, mode=mode).cpu(), grid)
    dec = vox2seq.decode(torch.arange(256 ** 3, dtype=torch.int32, device=cuda), mode=mode)
    assert np.array_equal(dec.cpu().numpy(), oracle_lib.vox2seq_decode(np.arange(256 ** 3), mode))
    # permute argument (serialized_attn.py:62-75 transposes x/y for the *_TRANSPOSE modes)
    p = vox2seq.encode(coords, permute=[1, 0, 2], mode=mode)
    assert torch.equal(p, vox2seq.encode(coords[:, [1, 0, 2]], mode=mode))
    assert torch.equal(vox2seq.decode(p, permute=[1, 0, 2], mode=mode), coords)
    assert vox2seq.encode(coords[:0], mode=mode).numel() == 0
