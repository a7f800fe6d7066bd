"""The temporal section of the row-block launch (csrc/rowblock.hip, gvf_rowblock_args.t_*): ONE launch that closes the spatial attention,
runs the block's temporal self attention on 48 / T tokens x T frames in registers and opens the image attention, against the three
launches it replaces (row-block launch | gvf_attn_fwd | row-block launch) -- same operands, same rounding points."""
import math

import pytest
import torch

from gvfdiffusion_amd import _lib
from gvfdiffusion_amd.ops import dit_ops

C, H = 512, 16


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _case(cuda, B, T, N, lp, seed, rms=True, adaln=True):
    g = torch.Generator().manual_seed(seed)
    M = B * T * N

    def r(*shape, s=1.0):
        return torch.randn(shape, generator=g) * s

    d = dict(M=M, T=T, N=N, B=B, lp=lp)
    d["a0"] = r(M, C).to(lp).to(cuda)
    d["w1"] = r(C, C, s=1 / math.sqrt(C)).to(lp).to(cuda)
    d["b1"] = r(C, s=0.1).to(cuda)
    d["x0"] = (r(M, C) * 2 + 0.5).to(cuda)
    d["mod"] = r(B, 9 * C, s=0.3).to(cuda)
    d["wqkv"] = r(3 * C, C, s=1.5 / math.sqrt(C)).to(lp).to(cuda)
    d["bqkv"] = r(3 * C, s=0.1).to(cuda)
    d["wout"] = r(C, C, s=1 / math.sqrt(C)).to(lp).to(cuda)
    d["bout"] = r(C, s=0.1).to(cuda)
    d["gq"] = (1 + 0.2 * r(C)).to(cuda) if rms else None
    d["gk"] = (1 + 0.2 * r(C)).to(cuda) if rms else None
    d["lw"], d["lb"] = (1 + 0.1 * r(C)).to(cuda), (0.1 * r(C)).to(cuda)
    d["w3"] = r(C, C, s=1 / math.sqrt(C)).to(lp).to(cuda)
    d["b3"] = r(C, s=0.1).to(cuda)
    d["adaln"] = adaln
    return d


def _chain(d, cuda):
    """row-block launch (to_out + adaLN + to_qkv) | attention over the frames | row-block launch (to_out + norm3 + to_q)"""
    M, T, N, B, lp, mod = d["M"], d["T"], d["N"], d["B"], d["lp"], d["mod"]
    TN, ld = T * N, 9 * C
    x = d["x0"].clone()
    qkv = torch.empty((M, 3 * C), dtype=lp, device=cuda)
    s2 = dit_ops.rowblock_pack_stream(d["w1"], w3=d["wqkv"])
    ln1 = dict(shift=mod[:, C:], scale=mod[:, 2 * C:]) if d["adaln"] else dict(ln_w=d["lw"], ln_b=d["lb"])
    dit_ops.rowblock_fused(d["a0"], s2, x, b1=d["b1"], gate1=mod[:, 0:] if d["adaln"] else None, ln1=ln1, mod_ld=ld, rows_per_group=TN,
                           out3=qkv, b3=d["bqkv"])
    ab = torch.empty((M, C), dtype=lp, device=cuda)
    st = (TN * 3 * C, 3 * C, N * 3 * C)
    dit_ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], ab, B, N, T, T, H, st, st, st, (TN * C, C, N * C), d["gq"], d["gk"])
    s3 = dit_ops.rowblock_pack_stream(d["wout"], w3=d["w3"])
    q = torch.empty((M, C), dtype=lp, device=cuda)
    dit_ops.rowblock_fused(ab, s3, x, b1=d["bout"], gate1=mod[:, 3 * C:] if d["adaln"] else None, ln1=dict(ln_w=d["lw"], ln_b=d["lb"]), mod_ld=ld,
                           rows_per_group=TN, out3=q, b3=d["b3"])
    return x, q, ab


def _merged(d, cuda):
    M, T, N, lp, mod = d["M"], d["T"], d["N"], d["lp"], d["mod"]
    ld = 9 * C
    x = d["x0"].clone()
    stream = dit_ops.rowblock_pack_stream(d["w1"], temporal=(d["wqkv"], d["wout"]), w3=d["w3"])
    q = torch.full((M, C), float("nan"), dtype=lp, device=cuda)
    ln1 = dict(shift=mod[:, C:], scale=mod[:, 2 * C:]) if d["adaln"] else dict(ln_w=d["lw"], ln_b=d["lb"])
    dit_ops.rowblock_fused(d["a0"], stream, x, b1=d["b1"], gate1=mod[:, 0:] if d["adaln"] else None, ln1=ln1, mod_ld=ld, rows_per_group=T * N,
                           out3=q, b3=d["b3"],
                           temporal=dict(frames=T, stride=N, b_qkv=d["bqkv"], gamma_q=d["gq"], gamma_k=d["gk"], b_out=d["bout"],
                                         gate=mod[:, 3 * C:] if d["adaln"] else None, ln=dict(ln_w=d["lw"], ln_b=d["lb"])))
    return x, q


@pytest.mark.gpu
@pytest.mark.parametrize("lp", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,N,rms,adaln", [(1, 24, 512, True, True), (2, 24, 64, True, True), (3, 12, 16, False, False), (1, 48, 7, True, False),
                                             (2, 16, 9, True, True), (1, 1, 96, True, True), (2, 8, 30, False, True)])
def test_temporal_section_equals_the_three_launches(cuda, lp, B, T, N, rms, adaln):
    d = _case(cuda, B, T, N, lp, seed=1000 * T + N, rms=rms, adaln=adaln)
    x_ref, q_ref, _ = _chain(d, cuda)
    x, q = _merged(d, cuda)
    rx, rq = rel_l2(x, x_ref), rel_l2(q, q_ref)
    print(f"temporal section {lp} B{B} T{T} N{N}: stream rel_l2 {rx:.2e}, projection rel_l2 {rq:.2e}")
    assert torch.isfinite(q.float()).all()
    # a 16-bit operand (attention output, normalised row) on a rounding boundary moves a stream element by one 16-bit ulp of the operand
    assert rx < (2e-4 if lp == torch.bfloat16 else 3e-5)
    assert rq < (3e-3 if lp == torch.bfloat16 else 4e-4)


@pytest.mark.gpu
def test_temporal_section_against_fp32_math(cuda):
    """fp16 operands against the same computation in fp32 torch (no 16-bit intermediate anywhere): bounds the section's own error."""
    lp = torch.float16
    d = _case(cuda, 2, 24, 32, lp, seed=7)
    B, T, N, M, mod = d["B"], d["T"], d["N"], d["M"], d["mod"]
    f = lambda t: t.float()
    rows = lambda v: v.reshape(B, 1, C).expand(B, T * N, C).reshape(M, C)
    x = d["x0"] + rows(mod[:, 0:C]) * (f(d["a0"]) @ f(d["w1"]).t() + d["b1"])
    ln = lambda v: torch.nn.functional.layer_norm(v, (C,), eps=1e-6)
    hb = ln(x) * (1 + rows(mod[:, 2 * C:3 * C])) + rows(mod[:, C:2 * C])
    qkv = (hb @ f(d["wqkv"]).t() + d["bqkv"]).reshape(B, T, N, 3, H, 32)
    q, k, v = (qkv[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3))          # (B, N, H, T, 32)
    nrm = lambda t, gam: torch.nn.functional.normalize(t, dim=-1) * gam.reshape(1, 1, H, 1, 32) * math.sqrt(32)
    q, k = nrm(q, d["gq"]), nrm(k, d["gk"])
    o = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(32), dim=-1) @ v        # (B, N, H, T, 32)
    o = o.permute(0, 3, 1, 2, 4).reshape(M, C)
    x = x + rows(mod[:, 3 * C:4 * C]) * (o @ f(d["wout"]).t() + d["bout"])
    q_ref = (ln(x) * d["lw"] + d["lb"]) @ f(d["w3"]).t() + d["b3"]
    x_new, q_new = _merged(d, cuda)
    rx, rq = rel_l2(x_new, x), rel_l2(q_new, q_ref)
    print(f"temporal section fp16 vs fp32 math: stream {rx:.2e}, projection {rq:.2e}")
    assert rx < 2e-4 and rq < 1.5e-3


@pytest.mark.gpu
def test_temporal_section_refuses_what_it_cannot_tile(cuda):
    d = _case(cuda, 1, 24, 64, torch.bfloat16, seed=3)
    stream = dit_ops.rowblock_pack_stream(d["w1"], temporal=(d["wqkv"], d["wout"]), w3=d["w3"])
    q = torch.empty((d["M"], C), dtype=torch.bfloat16, device=cuda)
    t = dict(frames=24, stride=64, b_qkv=d["bqkv"], gamma_q=d["gq"], gamma_k=d["gk"], b_out=d["bout"], ln=dict(ln_w=d["lw"], ln_b=d["lb"]))
    for bad, rpg in ((dict(t, frames=5), 24 * 64), (dict(t, stride=62), 24 * 64), (dict(t, gamma_k=None), 24 * 64), (t, 24 * 64 + 48)):
        with pytest.raises(_lib.GvfError):
            dit_ops.rowblock_fused(d["a0"], stream, d["x0"].clone(), b1=d["b1"], ln1=dict(ln_w=d["lw"], ln_b=d["lb"]),
                                   rows_per_group=rpg, out3=q, b3=d["b3"], temporal=bad)


@pytest.mark.gpu
@pytest.mark.parametrize("lp", ["bf16", "fp16"])
@pytest.mark.parametrize("B,T,N,fused", [(2, 16, 96, True), (1, 24, 128, True), (1, 6, 64, True), (2, 16, 64, True), (1, 5, 96, False), (1, 32, 64, False)])
def test_dit_uses_the_temporal_section_where_it_tiles_and_matches_the_three_launch_path(cuda, lp, B, T, N, fused):
    """DiT._blocks_rowblock: frames that divide 48 (and tokens that fill whole blocks) take the merged launch -- one launch with `temporal`
    per block and no gvf_attn_fwd -- everything else keeps the three launches; both give the same output."""
    import json
    import os
    from gvfdiffusion_amd import synthetic
    from gvfdiffusion_amd.model.dit import DiT
    man = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dit_manifest.json")))
    cfg = dict(man["config"], num_blocks=2)
    torch.manual_seed(T * 100 + N)
    net = DiT(**cfg)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():                 # upstream zero-initialises the gates: give every path a signal
            if p_.dim() >= 2:
                p_.copy_(torch.randn_like(p_) / math.sqrt(p_.shape[-1]))
            elif "gamma" in n_ or ("norm" in n_ and "weight" in n_):
                p_.copy_(1 + 0.1 * torch.randn_like(p_))
            else:
                p_.copy_(0.1 * torch.randn_like(p_))
    net = net.to(cuda).eval().set_compute_dtype(lp)
    inp = {k: v.to(cuda) for k, v in synthetic.dit_inputs(B=B, T=T, N=N, L_img=70, L_static=130, seed=5).items()}
    kw = dict(cond_images=inp["cond_images"], static_latent=inp["static_latent"], deformation_position_xyz=inp["deformation_position_xyz"])
    t = inp["t"] * torch.linspace(0.4, 1.0, B, device=cuda)
    calls = {"temporal": 0, "attn": 0}
    orig_f, orig_a = dit_ops.rowblock_fused, dit_ops.attention
    dit_ops.rowblock_fused = lambda *a, **k: (calls.__setitem__("temporal", calls["temporal"] + (k.get("temporal") is not None)), orig_f(*a, **k))[1]
    dit_ops.attention = lambda *a, **k: (calls.__setitem__("attn", calls["attn"] + 1), orig_a(*a, **k))[1]
    try:
        assert net.rowblock_temporal and net.use_rowblock
        y1 = net(inp["x"], t, **kw)
        n_fused, n_attn = calls["temporal"], calls["attn"]
        net.rowblock_temporal = False
        y0 = net(inp["x"], t, **kw)
    finally:
        dit_ops.rowblock_fused, dit_ops.attention = orig_f, orig_a
    assert (n_fused, n_attn) == ((2, 0) if fused else (0, 2)) and calls["attn"] == n_attn + 2
    r = rel_l2(y1, y0)
    print(f"DiT {lp} B{B} T{T} N{N}: merged temporal launch vs three launches {r:.2e}")
    assert r < (1e-6 if not fused else 2e-3 if lp == "bf16" else 2.5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("lp", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,N", [(2, 16, 64), (1, 4, 512), (3, 8, 100)])
def test_temporal_section_on_padded_groups(cuda, lp, B, T, N):
    """T * N tokens that do not fill whole 48-row blocks: a group (sample) owns ceil(N / (48 / T)) blocks, the rows behind its tokens are
    padding = the phantom tokens of its last block (the layout DiT._blocks_rowblock uses for T = 16 at N = 512).  Token rows of the merged
    launch against the three launches on the same padded buffers; the padding rows stay finite."""
    tpb = 48 // T
    TN, TNp = T * N, (N + tpb - 1) // tpb * 48
    assert TNp > TN and TNp == dit_ops.rowblock_padded_rows(TN)
    d = _case(cuda, B, T, N, lp, seed=17 * T + N)
    M = B * TNp

    def pad(t):                                         # (B * TN, c) -> (B * TNp, c), zeros behind every group's tokens
        out = torch.zeros((B, TNp, t.shape[1]), dtype=t.dtype, device=t.device)
        out[:, :TN] = t.view(B, TN, -1)
        return out.view(M, -1)
    a0, x0, mod = pad(d["a0"]), pad(d["x0"]), d["mod"]
    ld = 9 * C
    ln1, n3 = dict(shift=mod[:, C:], scale=mod[:, 2 * C:]), dict(ln_w=d["lw"], ln_b=d["lb"])
    # three launches
    x_ref = x0.clone()
    qkv = torch.zeros((M, 3 * C), dtype=lp, device=cuda)
    dit_ops.rowblock_fused(a0, dit_ops.rowblock_pack_stream(d["w1"], w3=d["wqkv"]), x_ref, b1=d["b1"], gate1=mod[:, 0:], ln1=ln1, mod_ld=ld,
                           rows_per_group=TNp, out3=qkv, b3=d["bqkv"])
    ab = torch.zeros((M, C), dtype=lp, device=cuda)
    st = (TNp * 3 * C, 3 * C, N * 3 * C)
    dit_ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], ab, B, N, T, T, H, st, st, st, (TNp * C, C, N * C), d["gq"], d["gk"])
    q_ref = torch.empty((M, C), dtype=lp, device=cuda)
    dit_ops.rowblock_fused(ab, dit_ops.rowblock_pack_stream(d["wout"], w3=d["w3"]), x_ref, b1=d["bout"], gate1=mod[:, 3 * C:], ln1=n3, mod_ld=ld,
                           rows_per_group=TNp, out3=q_ref, b3=d["b3"])
    # merged
    x = x0.clone()
    q = torch.full((M, C), float("nan"), dtype=lp, device=cuda)
    dit_ops.rowblock_fused(a0, dit_ops.rowblock_pack_stream(d["w1"], temporal=(d["wqkv"], d["wout"]), w3=d["w3"]), x, b1=d["b1"], gate1=mod[:, 0:],
                           ln1=ln1, mod_ld=ld, rows_per_group=TNp, out3=q, b3=d["b3"],
                           temporal=dict(frames=T, stride=N, b_qkv=d["bqkv"], gamma_q=d["gq"], gamma_k=d["gk"], b_out=d["bout"], gate=mod[:, 3 * C:], ln=n3))
    tok = lambda t: t.view(B, TNp, -1)[:, :TN]
    rx, rq = rel_l2(tok(x), tok(x_ref)), rel_l2(tok(q), tok(q_ref))
    print(f"temporal section, padded groups {lp} B{B} T{T} N{N} ({TNp - TN} padding rows): stream {rx:.2e}, projection {rq:.2e}")
    assert torch.isfinite(q.float()).all() and torch.isfinite(x).all()
    assert rx < (2e-4 if lp == torch.bfloat16 else 3e-5) and rq < (3e-3 if lp == torch.bfloat16 else 4e-4)
