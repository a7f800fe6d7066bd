"""Two samples in flight leave their serial results in the last bits (scripts/inflight_capture_repro.py), and the first evaluation of the
denoiser that differs sits where the OTHER slot has left its sampler.  Which of the other slot's stages does it?  Thread A runs the same
48 DiT evaluations every round (eager launches, fp16) and compares one word per evaluation with the serial run; thread B loops ONE kind of
work on its own stream meanwhile: motion-VAE decode | batched render | farthest point sampling | prepare_conditions (library GEMMs + K / V^T
pack) | another DiT | nothing.

    python scripts/inflight_which_stage.py
"""
import json
import os
import sys
import threading

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from gvfdiffusion_amd import synthetic
    from gvfdiffusion_amd.model.dit import DiT
    from gvfdiffusion_amd.model.autoencoder import GSKLTemporalVariationalAutoEncoder
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from gvfdiffusion_amd.utils import sample_gs
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "dit_manifest.json")))
    vman = json.load(open(os.path.join(ROOT, "tests", "golden", "vae_manifest.json")))
    T, P = 24, 32768
    dits = []
    for _ in range(2):
        m = DiT(**man["config"])
        m.load_state_dict(synthetic.dit_state_dict(man["state_dict"], seed=0), strict=True)
        dits.append(m.to(dev).eval())
    g = torch.Generator().manual_seed(5)
    cond = [(torch.randn((1, T, 1370, 1024), generator=g).to(dev), torch.randn((1, 4096, 14), generator=g).to(dev), torch.rand((1, 512, 3), generator=g).to(dev))
            for _ in range(2)]
    x = torch.randn((1, T, 512, 16), generator=g).to(dev)
    ts = [torch.tensor([1000.0 * (1 - k / 50)]).to(dev) for k in range(48)]
    torch.manual_seed(0)
    vae = GSKLTemporalVariationalAutoEncoder(**vman["config"], num_timesteps=T)
    with torch.no_grad():
        for p_ in vae.parameters():
            p_.copy_(torch.randn_like(p_) * (1.0 / p_.shape[1] ** 0.5 if p_.dim() == 2 else 0.05))
    vae = vae.to(dev).eval()
    attrs = synthetic.random_gaussians(P, sh_degree=0, seed=3)
    gm = synthetic.gaussian_model_from(attrs, 0, dev)
    queries = torch.cat([gm._xyz, gm._features_dc.reshape(P, 3), gm._scaling, gm._rotation, gm._opacity], 1)[None].float()
    lat = torch.randn((T, 512, 16), generator=g).to(dev)
    rend = GaussianRenderer({"resolution": 256, "near": synthetic.NEAR, "far": synthetic.FAR, "ssaa": 1, "bg_color": (1, 1, 1)})
    rend.pipe.use_mip_gaussian = True
    rend.pipe.kernel_size = synthetic.KERNEL_2D
    ext = torch.stack([synthetic.orbit_w2c(360.0 * f / T, 15.0) for f in range(T)]).to(dev)
    K = synthetic.intrinsics().to(dev)
    delta = (torch.randn((T, P, 14), generator=g) * 0.01).to(dev)
    gs = torch.cat([gm.get_xyz, gm._features_dc.reshape(-1, 3), gm.get_opacity.reshape(-1, 1), gm.get_scaling, gm.get_rotation], 1).float()

    # REPRO_TRACE_OPS=1: one word per OPERATOR call of the forward (the outputs of every dit_ops call that writes a tensor), to name the launch
    # whose result moves first
    op_trace, op_names = [], []
    if os.environ.get("REPRO_TRACE_OPS") == "1":
        from gvfdiffusion_amd.ops import dit_ops as D_

        def wrap(name, fn, outs):
            def w(*a, **k):
                r = fn(*a, **k)
                if threading.current_thread() is trace_thread[0]:
                    for tag, tsr in outs(a, k, r):
                        if tsr is not None:
                            op_trace.append(tsr.reshape(-1).view(torch.uint8)[: tsr.numel() * tsr.element_size() // 4 * 4].view(torch.int32).sum(dtype=torch.int64))
                            op_names.append(name + ":" + tag)
                return r
            return w
        D_.attention_tiled = wrap("attention_tiled", D_.attention_tiled, lambda a, k, r: [("out", a[3])])
        D_.rowblock_fused = wrap("rowblock_fused", D_.rowblock_fused, lambda a, k, r: [("x", a[2]), ("out3", k.get("out3")), ("hb_out", k.get("hb_out")),
                                                                                      ("kt", (k.get("kv_tiles") or (None, None))[0]), ("vt", (k.get("kv_tiles") or (None, None))[1])])
        D_.final_layer_f32 = wrap("final_layer_f32", D_.final_layer_f32, lambda a, k, r: [("y", a[3])])
        D_.timestep_embed_f32 = wrap("timestep_embed_f32", D_.timestep_embed_f32, lambda a, k, r: [("s2", r)])
        D_.modulation_f32 = wrap("modulation_f32", D_.modulation_f32, lambda a, k, r: [("mod", r)])
    trace_thread = [threading.current_thread()]

    def chain_a():
        out = []
        del op_trace[:]; del op_names[:]
        trace_thread[0] = threading.current_thread()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            for t in ts:
                y = dits[0](x, t, *cond[0])
                out.append(y.float().view(torch.int32).sum(dtype=torch.int64))
        return torch.stack(out)

    def work_b(kind):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            if kind == "vae_decode":
                vae.decode(lat, queries)
            elif kind == "render":
                rend.render_frames(gm, ext, K, delta_pc=delta, sync=False)
            elif kind == "fps":
                sample_gs([gs], num_latents=4096, device=dev, random_start=False)
            elif kind == "prepare_conditions":
                c = tuple(t_.clone() for t_ in cond[1])
                dits[1].prepare_conditions(c[0], c[1], c[2], T)
            elif kind == "dit":
                dits[1](x, ts[3], *cond[1])
            elif kind == "copy":
                lat.clone()
            elif kind == "vae_latents":               # the decode's first half: latent self-attention / GEGLU blocks (gemm.hip, attn.hip, vae.hip)
                vae.decode_latents(lat)
            elif kind == "kvres64":                   # head_dim-64 cross attention of the decode alone
                from gvfdiffusion_amd.ops import dit_ops
                H_, d_, L_, n_ = 8, 64, 512, 16384
                if not hasattr(work_b, "kv"):
                    gg = torch.Generator().manual_seed(9)
                    work_b.kv = (torch.randn((T, n_, H_ * d_), generator=gg).to(dev).half(), torch.randn((T, L_, H_ * d_), generator=gg).to(dev).half(),
                                 torch.randn((T, L_, H_ * d_), generator=gg).to(dev).half(), torch.empty((T, n_, H_ * d_), dtype=torch.float16, device=dev))
                q_, k_, v_, o_ = work_b.kv
                sq, sk = (n_ * H_ * d_, 0, H_ * d_), (L_ * H_ * d_, 0, H_ * d_)
                dit_ops.attention(q_, k_, v_, o_, T, 1, n_, L_, H_, sq, sk, sk, sq, None, None, head_dim=d_)
            elif kind == "gemm":
                from gvfdiffusion_amd.ops import dit_ops
                if not hasattr(work_b, "gm"):
                    gg = torch.Generator().manual_seed(10)
                    work_b.gm = (torch.randn((32768, 512), generator=gg).to(dev).half(), torch.randn((2048, 512), generator=gg).to(dev).half(),
                                 torch.empty((32768, 2048), dtype=torch.float16, device=dev))
                a_, w_, o_ = work_b.gm
                dit_ops.gemm(a_, w_, None, o_, dit_ops.EPI_STORE_16)

    kinds = os.environ.get("REPRO_KINDS", "nothing,copy,dit,prepare_conditions,fps,render,vae_decode").split(",")
    ref = chain_a()
    ref_ops = torch.stack(op_trace) if op_trace else None
    ref_names = list(op_names)
    for k in kinds:
        if k != "nothing":
            work_b(k)
    torch.cuda.synchronize()
    again = chain_a()
    print(json.dumps({"serial repeat identical": bool(torch.equal(ref, again))}), flush=True)
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    R = int(os.environ.get("REPRO_ROUNDS", "6"))
    for kind in kinds:
        bad, first = 0, []
        for r in range(R):
            stop = threading.Event()
            res = [None]

            def a():
                torch.cuda.set_device(dev)
                with torch.cuda.stream(sa):
                    res[0] = chain_a()
                    sa.synchronize()
                stop.set()

            def b():
                torch.cuda.set_device(dev)
                with torch.cuda.stream(sb):
                    while not stop.is_set():
                        if kind != "nothing":
                            work_b(kind)
                        sb.synchronize()
            th = [threading.Thread(target=a), threading.Thread(target=b)]
            [t.start() for t in th]; [t.join() for t in th]
            torch.cuda.synchronize()
            ne = res[0] != ref
            if bool(ne.any()):
                bad += 1
                first.append(int(ne.nonzero()[0]))
                if ref_ops is not None:
                    cur = torch.stack(op_trace)
                    dn = (cur != ref_ops).nonzero().reshape(-1).tolist()
                    per = len(ref_names) // len(ts)
                    print(json.dumps({"first divergent operator outputs": [(i // per, i % per, ref_names[i]) for i in dn[:6]], "operator outputs per evaluation": per,
                                      "divergent": len(dn)}), flush=True)
        print(json.dumps({"other slot runs": kind, "rounds": R, "divergent_rounds": bad, "first_divergent_evaluation": first,
                          "evaluations_differing_last_round": int(ne.sum())}), flush=True)


if __name__ == "__main__":
    main()
