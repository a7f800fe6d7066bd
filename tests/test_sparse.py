"""Sparse side of the path: SparseTensor container, index builders against goldens made
by the reference's own calc_window_partition / calc_serialization, and (GPU) the varlen / windowed / serialized
attention operators against dense per-sequence attention."""
import os

import math
import numpy as np
import pytest
import torch

from gvfdiffusion_amd.sparse import SparseTensor, sparse_cat, sparse_unbind, sparse_batch_broadcast

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "sparse_index_golden.npz"))


def _tensor(dev="cpu", C=8, seed=0):
    coords = torch.from_numpy(G["coords"]).to(dev)
    feats = torch.randn((coords.shape[0], C), generator=torch.Generator().manual_seed(seed)).to(dev)
    return SparseTensor(feats, coords)


def test_container_surface():
    t = _tensor()
    assert t.shape == torch.Size([3, 8]) and [s.stop - s.start for s in t.layout] == [300, 77, 513]
    u = t.replace(t.feats * 2)
    assert u.layout is t.layout and torch.equal((t + t).feats, u.feats) and torch.equal((2 * t).feats, u.feats)
    per_batch = torch.arange(3.0).reshape(3, 1).expand(3, 8)
    bb = sparse_batch_broadcast(t, per_batch)
    assert float(bb[0, 0]) == 0 and float(bb[300, 0]) == 1 and float(bb[-1, 0]) == 2
    assert torch.equal((t + per_batch).feats, t.feats + bb)
    one = t[1]
    assert one.shape[0] == 1 and one.feats.shape[0] == 77 and int(one.coords[:, 0].max()) == 0
    cat = sparse_cat([t[0], t[2]])
    assert cat.shape[0] == 2 and cat.feats.shape[0] == 813 and int(cat.coords[-1, 0]) == 1
    parts = sparse_unbind(t, 0)
    assert len(parts) == 3 and torch.equal(parts[2].feats, t.feats[t.layout[2]])
    r = t.reshape(2, 4)
    assert r.shape == torch.Size([3, 2, 4]) and len(r.unbind(1)) == 2
    d = t.dense()
    c = t.coords[5].long()
    assert d.shape[:2] == (3, 8) and torch.equal(d[c[0], :, c[1], c[2], c[3]], t.feats[5])
    t.register_spatial_cache("k", 123)
    assert t.get_spatial_cache("k") == 123 and t.replace(t.feats).get_spatial_cache("k") == 123
    assert t.half().dtype == torch.float16 and t.float().dtype == torch.float32
    f = SparseTensor.full([0, 0, 0, 1, 1, 1], (2, 3), 0.5)
    assert f.feats.shape == (16, 3) and f.shape[0] == 2


def test_window_partition_matches_reference():
    from gvfdiffusion_amd.sparse.attention import calc_window_partition
    t = _tensor()
    for key, ws, sh in (("win_8_0", 8, 0), ("win_8_4", 8, 4), ("win_5_1_2_3", 5, (1, 2, 3))):
        fwd, bwd, lens, bidx = calc_window_partition(t, ws, sh)
        assert lens == G[key + "_lens"].tolist() and bidx == G[key + "_bidx"].tolist()
        assert torch.equal(fwd[bwd], torch.arange(fwd.shape[0]))                 # bwd is the inverse permutation
        gf = torch.from_numpy(G[key + "_fwd"])
        off = 0
        for n in lens:                                                            # same members per window (order inside a
            assert torch.equal(fwd[off:off + n].sort().values, gf[off:off + n].sort().values)   # window is unspecified upstream)
            off += n


@pytest.mark.gpu
def test_serialization_matches_reference(cuda):
    from gvfdiffusion_amd.sparse.attention import calc_serialization, SerializeMode
    t = _tensor(cuda)
    for mode in SerializeMode:
        for ws, ss, sw in ((32, 0, (0, 0, 0)), (48, 7, (3, 0, 5))):
            fwd, bwd, lens, bidx = calc_serialization(t, ws, mode, ss, sw)
            key = f"ser_{mode.name}_{ws}_{ss}"
            assert lens == G[key + "_lens"].tolist() and bidx == G[key + "_bidx"].tolist()
            assert np.array_equal(fwd.cpu().numpy(), G[key + "_fwd"]), key       # codes are unique per sample: exact
            assert np.array_equal(bwd.cpu().numpy(), G[key + "_bwd"]), key


def _dense_ref(q, k, v):
    """[L,H,C] single-sequence attention in fp32 on bf16-rounded inputs, bf16-rounded P (the kernel's rounding model)."""
    from oracle import dit_ref
    return dit_ref.sdpa(q[None].float(), k[None].float(), v[None].float(), "bf16")[0]


@pytest.mark.gpu
@pytest.mark.parametrize("C", [32, 64])
def test_varlen_attention_all_call_forms(cuda, C):
    from gvfdiffusion_amd.sparse.attention import sparse_scaled_dot_product_attention as spa
    H = 3
    g = torch.Generator().manual_seed(C)
    coords = torch.from_numpy(G["coords"]).to(cuda)
    T = coords.shape[0]
    qkv = SparseTensor(torch.randn((T, 3, H, C), generator=g).to(cuda).to(torch.bfloat16), coords)
    out = spa(qkv)
    assert isinstance(out, SparseTensor) and out.feats.shape == (T, H, C)
    for sl in qkv.layout:
        f = qkv.feats[sl]
        ref = _dense_ref(f[:, 0], f[:, 1], f[:, 2])
        assert float((out.feats[sl].float() - ref).norm() / ref.norm()) < 6e-3
    q, k, v = qkv.unbind(1)
    assert torch.equal(spa(q, k, v).feats, out.feats)
    kv = qkv.replace(qkv.feats[:, 1:])
    assert torch.equal(spa(q, kv).feats, out.feats)
    # sparse q against a dense context (cross attention form) and dense q against sparse kv
    ctx = torch.randn((3, 50, 2, H, C), generator=g).to(cuda).to(torch.bfloat16)
    o2 = spa(q, ctx)
    for b, sl in enumerate(q.layout):
        ref = _dense_ref(q.feats[sl], ctx[b, :, 0], ctx[b, :, 1])
        assert float((o2.feats[sl].float() - ref).norm() / ref.norm()) < 6e-3
    qd = torch.randn((3, 40, H, C), generator=g).to(cuda).to(torch.bfloat16)
    o3 = spa(qd, kv)
    assert o3.shape == (3, 40, H, C)
    for b, sl in enumerate(kv.layout):
        ref = _dense_ref(qd[b], kv.feats[sl][:, 0], kv.feats[sl][:, 1])
        assert float((o3[b].float() - ref).norm() / ref.norm()) < 6e-3
    # fp16 inputs (what the reference's fp16 torso / autocast hands flash-attn) are contracted in fp16, not down-cast to bf16: 8 x closer
    qkv16 = qkv.replace(qkv.feats.to(torch.float16))
    o16 = spa(qkv16)
    assert o16.feats.dtype == torch.float16
    for sl in qkv16.layout:
        f = qkv16.feats[sl].float()
        ref = (torch.softmax(torch.einsum("qhc,khc->hqk", f[:, 0], f[:, 1]) / math.sqrt(C), dim=-1) @ f[:, 2].permute(1, 0, 2)).permute(1, 0, 2)
        assert float((o16.feats[sl].float() - ref).norm() / ref.norm()) < 8e-4


@pytest.mark.gpu
def test_windowed_and_serialized_attention(cuda):
    from gvfdiffusion_amd.sparse.attention import (sparse_windowed_scaled_dot_product_self_attention as win_attn,
                                                   sparse_serialized_scaled_dot_product_self_attention as ser_attn,
                                                   calc_window_partition, calc_serialization, SerializeMode)
    H, C = 2, 64
    coords = torch.from_numpy(G["coords"]).to(cuda)
    qkv = SparseTensor(torch.randn((coords.shape[0], 3, H, C), generator=torch.Generator().manual_seed(1)).to(cuda).to(torch.bfloat16), coords)
    out = win_attn(qkv, 8, (4, 4, 4))
    fwd, bwd, lens, _ = calc_window_partition(qkv, 8, (4, 4, 4))
    f = qkv.feats[fwd]
    off = 0
    for n in lens:
        ref = _dense_ref(f[off:off + n, 0], f[off:off + n, 1], f[off:off + n, 2])
        got = out.feats[fwd[off:off + n]]
        assert float((got.float() - ref).norm() / ref.norm()) < 6e-3
        off += n
    assert qkv.get_spatial_cache("window_partition_8_(4, 4, 4)") is not None     # cached on the tensor
    out = ser_attn(qkv, 32, SerializeMode.HILBERT, 0, (0, 0, 0))
    fwd, bwd, lens, _ = calc_serialization(qkv, 32, SerializeMode.HILBERT, 0, (0, 0, 0))
    f = qkv.feats[fwd]
    full = torch.cat([_dense_ref(f[o:o + n, 0], f[o:o + n, 1], f[o:o + n, 2]) for o, n in zip(np.cumsum([0] + lens[:-1]), lens)])
    ref = full[bwd]
    assert float((out.feats.float() - ref).norm() / ref.norm()) < 6e-3


@pytest.mark.gpu
def test_sparse_multi_head_attention_module(cuda):
    from gvfdiffusion_amd.sparse.attention import SparseMultiHeadAttention, SerializeMode
    torch.manual_seed(0)
    coords = torch.from_numpy(G["coords"]).to(cuda)
    x = SparseTensor(torch.randn((coords.shape[0], 128)).to(cuda), coords)
    for kw in (dict(attn_mode="full", qk_rms_norm=True), dict(attn_mode="windowed", window_size=8, shift_window=(0, 0, 0)),
               dict(attn_mode="serialized", window_size=32, serialize_mode=SerializeMode.Z_ORDER, shift_sequence=0, shift_window=(0, 0, 0))):
        m = SparseMultiHeadAttention(128, 4, **kw).to(cuda)
        y = m(x)
        assert isinstance(y, SparseTensor) and y.feats.shape == (coords.shape[0], 128) and torch.isfinite(y.feats).all()
        if kw["attn_mode"] == "full":           # reference semantics in fp32 torch
            qkv = torch.nn.functional.linear(x.feats, m.to_qkv.weight, m.to_qkv.bias).reshape(-1, 3, 4, 32)
            q = torch.nn.functional.normalize(qkv[:, 0], dim=-1) * m.q_rms_norm.gamma * 32 ** 0.5
            k = torch.nn.functional.normalize(qkv[:, 1], dim=-1) * m.k_rms_norm.gamma * 32 ** 0.5
            outs = []
            for sl in x.layout:
                a = torch.softmax(torch.einsum("qhc,khc->hqk", q[sl], k[sl]) / 32 ** 0.5, dim=-1)
                outs.append(torch.einsum("hqk,khc->qhc", a, qkv[sl, 2]).reshape(-1, 128))
            ref = torch.nn.functional.linear(torch.cat(outs), m.to_out.weight, m.to_out.bias)
            assert float((y.feats - ref).detach().norm() / ref.detach().norm()) < 2e-2
    # the older channel layout [head][q|k|v][c] (sparse/attention/modules.py:150-162) = the same attention with permuted rows
    new = SparseMultiHeadAttention(128, 4, attn_mode="windowed", window_size=8, shift_window=4, qk_rms_norm=True).to(cuda)
    old = SparseMultiHeadAttention(128, 4, attn_mode="windowed", window_size=8, shift_window=4, qk_rms_norm=True, use_old_attn_impl=True).to(cuda)
    with torch.no_grad():
        old.load_state_dict(new.state_dict())
        old.to_qkv.weight.copy_(new.to_qkv.weight.reshape(3, 4, 32, 128).permute(1, 0, 2, 3).reshape(384, 128))
        old.to_qkv.bias.copy_(new.to_qkv.bias.reshape(3, 4, 32).permute(1, 0, 2).reshape(384))
        assert float((old(x).feats - new(x).feats).abs().max()) < 1e-5
    cross = SparseMultiHeadAttention(128, 4, ctx_channels=64, type="cross").to(cuda)
    ctx = torch.randn((3, 20, 64)).to(cuda)
    assert cross(x, ctx).feats.shape == (coords.shape[0], 128)


def test_sparse_norm_and_spatial_layers_match_the_reference():
    """sparse/norm.py:12-27, sparse/spatial.py:13-110 on a ragged two-sample batch: outputs of the reference's own classes
    (tests/golden/make_golden.py::gen_sparse_layers).  Container-level torch plumbing: runs wherever the tensors live."""
    import os
    import numpy as np
    from gvfdiffusion_amd import sparse as sp
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sparse_layers_golden.npz"))
    x = sp.SparseTensor(torch.from_numpy(z["feats"]), torch.from_numpy(z["coords"]))
    gn = sp.SparseGroupNorm(3, 12)
    with torch.no_grad():
        gn.weight.copy_(torch.from_numpy(z["gn_w"])); gn.bias.copy_(torch.from_numpy(z["gn_b"]))
        assert np.abs(gn(x).feats.numpy() - z["gn_out"]).max() < 1e-5
        for tag, f in (("2", 2), ("211", (2, 1, 1))):
            d = sp.SparseDownsample(f)(x)
            assert np.array_equal(d.coords.numpy(), z[f"down{tag}_coords"]) and np.abs(d.feats.numpy() - z[f"down{tag}_feats"]).max() < 1e-6
            u = sp.SparseUpsample(f)(d)
            assert np.array_equal(u.coords.numpy(), z[f"up{tag}_coords"]) and np.abs(u.feats.numpy() - z[f"up{tag}_feats"]).max() < 1e-6
            assert u.layout == x.layout
        s = sp.SparseSubdivide()(x)
        assert np.array_equal(s.coords.numpy(), z["sub_coords"]) and np.array_equal(s.feats.numpy(), z["sub_feats"])
    with pytest.raises(ValueError):
        sp.SparseUpsample(4)(x)
