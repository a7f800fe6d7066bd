"""The hot-path half of the reference's inference_dpm_latent.py on the MI355X package: same flags, same chain, sharded over the ranks.

Reference main() (inference_dpm_latent.py:41-273), and what this script does with each stage:
  :65-66, 177-203  TRELLIS image -> canonical static Gaussians + azimuth alignment   OUT OF SCOPE (image -> 3D generator, needs downloaded
                   weights): the canonical Gaussians come from --static_gs (a .pt list of (P, 14) tensors [xyz3 | rgb3 | op1 | scale3 | rot4],
                   what get_gaussian_tensor returns, train_vae.py:466-472) or are synthetic (--synthetic)
  :74-116          DiT / motion VAE built from the config, checkpoints loaded with the "module." prefix stripped   same; --synthetic draws
                   deterministic weights of the released architectures instead (no checkpoint exists offline)
  :122-125, 142    accelerate, mixed_precision = fp16 if --use_fp16   one process per GPU (torch.distributed.run / accelerate launch set RANK /
                   WORLD_SIZE); --use_fp16 puts the DiT and the VAE decode inside torch.autocast(fp16), which the kernels follow
  :156, 225-249    NoiseScheduleVP -> model_wrapper (v-prediction, two-scale guidance) -> DPM_Solver.sample (adaptive | multistep)   same calls
  :208-222         farthest point sampling of 512 / 4096 Gaussians -> conditions   same (gvf_fps)
  :250-257         de-normalise, vae.decode -> (B, T, P, 14) deltas   same
  :261-272         render_and_save_images: 32 timesteps x 128 orbit cameras, 512 x 512 PNGs   rendered in batched launches on the device
                   (--views cameras per timestep); PNGs only with --save_png
  (new)            samples are SHARDED over the ranks (the reference renders the same samples on every rank) and the finished uint8 frames
                   gathered with ONE all-gather: gvfdiffusion_amd.distributed.

    python inference_dpm_latent.py --synthetic --num_samples 2 --use_fp16 --adaptive
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 inference_dpm_latent.py --synthetic --num_samples 8
"""
import argparse
import contextlib
import json
import os
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
MODEL_TYPES = {"eps": "noise", "xstart": "x_start", "v": "v"}          # inference_dpm_latent.py:34-38


def create_argparser():
    def none_or_str(value):
        return None if value.lower() == "none" else value
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    # the reference's flags (inference_dpm_latent.py:276-316) that concern the path from latents to frames
    p.add_argument("--exp_name", type=str, default="/tmp/output/")
    p.add_argument("--ckpt", type=str, default=None)
    p.add_argument("--vae_ckpt", type=str, default=None)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--use_fp16", action="store_true")
    p.add_argument("--config", type=str, default="configs/diffusion.yml")
    p.add_argument("--deformation_mean_file", type=none_or_str, default=None)
    p.add_argument("--deformation_std_file", type=none_or_str, default=None)
    p.add_argument("--static_mean_file", type=none_or_str, default=None)
    p.add_argument("--static_std_file", type=none_or_str, default=None)
    p.add_argument("--num_timesteps", type=int, default=24)
    p.add_argument("--num_samples", type=int, default=10)
    p.add_argument("--rescale_timesteps", type=int, default=100)
    p.add_argument("--guidance_scale", type=float, default=1.0)
    p.add_argument("--guidance_scale2", type=float, default=1.0)
    p.add_argument("--adaptive", action="store_true")
    # where the reference's upstream stages are replaced by files / synthetic inputs
    p.add_argument("--synthetic", action="store_true", help="random-init weights of the released architectures, synthetic Gaussians and conditions")
    p.add_argument("--static_gs", type=str, default=None, help=".pt file: list of (P, 14) canonical Gaussian tensors, one per sample")
    p.add_argument("--cond_images", type=str, default=None, help=".pt file: (num_samples, T, 1370, 1024) DINOv2 features")
    p.add_argument("--gaussians", type=int, default=32_768, help="--synthetic: Gaussians per sample")
    p.add_argument("--resolution", type=int, default=512, help="render size (the reference forces 512, inference_dpm_latent.py:161-162)")
    p.add_argument("--views", type=int, default=4, help="orbit cameras per timestep (the reference renders 128)")
    p.add_argument("--in_flight", type=int, default=1, help="samples of a rank in flight on separate HIP streams (one model copy each)")
    p.add_argument("--save_png", action="store_true")
    return p


def _load_config(path):
    """`model:` / `diffusion:` / `motion_vae:` sections of configs/diffusion.yml; offline fallback: the manifests the goldens were generated from."""
    if os.path.exists(path):
        import yaml
        cfg = yaml.safe_load(open(path))
        return cfg["model"], cfg.get("diffusion", {}), cfg["motion_vae"]
    dit = json.load(open(os.path.join(ROOT, "tests", "golden", "dit_manifest.json")))
    vae = json.load(open(os.path.join(ROOT, "tests", "golden", "vae_manifest.json")))
    return dit["config"], dict(noise_schedule="cosine", predict_type="v"), vae["config"]


def _strip_module(sd):
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}          # inference_dpm_latent.py:78-87


def build_models(args, dev, n_copies=1):
    from gvfdiffusion_amd import synthetic
    from gvfdiffusion_amd.model.autoencoder import GSKLTemporalVariationalAutoEncoder
    from gvfdiffusion_amd.model.dit import DiT
    model_cfg, diff_cfg, vae_cfg = _load_config(args.config)
    if not args.synthetic and (args.ckpt is None or args.vae_ckpt is None):
        raise SystemExit("give --ckpt and --vae_ckpt (released checkpoints) or --synthetic")
    copies = []
    for _ in range(n_copies):
        dit = DiT(**model_cfg)
        vae = GSKLTemporalVariationalAutoEncoder(**vae_cfg, num_timesteps=args.num_timesteps)
        if args.synthetic:
            man = json.load(open(os.path.join(ROOT, "tests", "golden", "dit_manifest.json")))
            dit.load_state_dict(synthetic.dit_state_dict(man["state_dict"], seed=0), strict=True)
            g = torch.Generator().manual_seed(1)
            with torch.no_grad():
                for prm in vae.parameters():
                    prm.copy_(torch.randn(prm.shape, generator=g) * (1.0 / prm.shape[1] ** 0.5 if prm.dim() == 2 else 0.05))
                vae.to_outputs.weight.mul_(0.02)
        else:
            dit.load_state_dict(_strip_module(torch.load(args.ckpt, map_location="cpu")), strict=True)
            vae.load_state_dict(_strip_module(torch.load(args.vae_ckpt, map_location="cpu")), strict=True)
        # Every forward of a sample replays ONE hipGraph (captured on the sample's first step: new conditions, new capture) -- also with several
        # samples in flight: captures are serialised and thread-local (DiT._forward_graphed).  Rounds 3-4 saw in-flight samples leave their
        # serial results in the last bits and blamed the capture; the cause was packed-fp32 arithmetic beside another wave's MFMAs
        # (profiles/r04_inflight_root_cause.txt), fixed in the build; GVF_INFLIGHT_GRAPH=0 restores eager launches for in-flight instances.
        copies.append((dit.to(dev).eval().enable_graph(n_copies == 1 or os.environ.get("GVF_INFLIGHT_GRAPH", "1") == "1"), vae.to(dev).eval()))
    return copies, model_cfg, diff_cfg, vae_cfg


def sample_inputs(args, i, dev, model_cfg):
    """Canonical Gaussians (P, 14) and DINOv2 conditions of global sample i: files if given, else seeded synthetic ones."""
    from gvfdiffusion_amd import synthetic
    T = args.num_timesteps
    if args.static_gs:
        gs = torch.load(args.static_gs, map_location="cpu")[i].float().to(dev)
    else:
        a = synthetic.random_gaussians(args.gaussians, sh_degree=0, seed=1000 + i)
        gm = synthetic.gaussian_model_from(a, 0, dev)
        gs = torch.cat([gm.get_xyz, gm._features_dc.reshape(-1, 3), gm.get_opacity.reshape(-1, 1), gm.get_scaling, gm.get_rotation], 1).float()
    if args.cond_images:
        cond = torch.load(args.cond_images, map_location="cpu")[i:i + 1].float().to(dev)
    else:
        cond = torch.randn((1, T, 1370, model_cfg["image_cond_channels"]), generator=torch.Generator().manual_seed(2000 + i)).to(dev)
    return gs, cond


def build_chain(args, dev, probe=None):
    """Models, schedule, renderers and cameras of one rank -> (chain, stats, n_fl): chain(slot, i) produces global sample i's finished frames
    on copy `slot` of the models; stats collects (sample, NFE, seconds).  probe: optional dict, filled with {i: (latents, deltas)} clones
    (scripts/inflight_capture_repro.py compares them between runs)."""
    from gvfdiffusion_amd import synthetic
    from gvfdiffusion_amd.model.dpmsolver import NoiseScheduleVP, model_wrapper, DPM_Solver
    from gvfdiffusion_amd.model.gaussian_diffusion import create_gaussian_diffusion
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from gvfdiffusion_amd.utils import orbit_cameras, pad_static_gs, render_sample_frames, sample_gs
    n_fl = max(1, args.in_flight)
    copies, model_cfg, diff_cfg, vae_cfg = build_models(args, dev, n_fl)
    diffusion = create_gaussian_diffusion(**{k: v for k, v in diff_cfg.items() if k in ("steps", "noise_schedule", "predict_type")})
    ns = NoiseScheduleVP("discrete", betas=torch.from_numpy(diffusion.betas))              # :156
    stat = lambda f, d: torch.load(f, map_location="cpu").float().to(dev) if f else torch.tensor(d, device=dev)     # noqa: E731
    d_mean, d_std = stat(args.deformation_mean_file, 0.0), stat(args.deformation_std_file, 1.0)
    s_mean, s_std = stat(args.static_mean_file, 0.0), stat(args.static_std_file, 1.0)
    rends = []                                            # one renderer per slot in flight (render_sample_frames toggles its pipe options)
    for _ in range(n_fl):
        r = GaussianRenderer({"resolution": args.resolution, "near": synthetic.NEAR, "far": synthetic.FAR, "ssaa": 1, "bg_color": (1, 1, 1)})
        r.pipe.kernel_size = synthetic.KERNEL_2D
        rends.append(r)
    K = synthetic.intrinsics().to(dev)
    cams = orbit_cameras(args.views).to(dev)
    T = args.num_timesteps
    autocast = (lambda: torch.autocast("cuda", dtype=torch.float16)) if args.use_fp16 else contextlib.nullcontext
    stats = []

    def chain(slot, i):
        """global sample i -> its finished frames (T * views, 3, S, S) uint8 on the device (inference_dpm_latent.py:208-272)."""
        dit, vae = copies[slot]
        t0 = time.perf_counter()
        gs, cond_images = sample_inputs(args, i, dev, model_cfg)
        # (start at the first Gaussian: the reference's random start, torch_cluster's default, would make a sample depend on how many samples
        # drew from the global generator before it -- i.e. on the sharding)
        fps512 = sample_gs([gs], num_latents=vae_cfg["num_latents"], device=dev, random_start=False)      # :208
        fps4096 = sample_gs([gs], num_latents=4096, device=dev, random_start=False)                       # :209
        padded, valid_idx = pad_static_gs([gs])                                                   # :210
        condition = {"cond_images": cond_images, "static_latent": (fps4096 - s_mean) / s_std, "deformation_position_xyz": fps512[..., :3].contiguous()}
        uncond = dict(condition, cond_images=torch.zeros_like(cond_images))
        fn = model_wrapper(dit, ns, model_type=MODEL_TYPES[diff_cfg.get("predict_type", "v")], model_kwargs={}, guidance_type="classifier-free",
                           guidance_scale=args.guidance_scale, guidance_scale2=args.guidance_scale2, condition=condition,
                           unconditional_condition=uncond)
        nfe = {"n": 0}
        trace = []

        def counted(x, t):
            nfe["n"] += 1
            y = fn(x, t)
            if probe is not None:                              # one word per evaluation: where does a run leave its reference?
                trace.append(y.detach().float().view(torch.int32).sum(dtype=torch.int64))
            return y
        counted.sampling_scope = fn.sampling_scope      # ... and brackets each sample() call: the DiT checks its weights once per sample
        counted.prepare_times = fn.prepare_times          # the solver announces its time grid through the wrapper it is handed (modulation table, one upload)
        noise = torch.randn((1, T, model_cfg["resolution"], model_cfg["in_channels"]), generator=torch.Generator().manual_seed(args.seed + i)).to(dev)
        solver = DPM_Solver(counted, ns, algorithm_type="dpmsolver++")
        solver.verbose = False          # (not contextlib.redirect_stdout: chain() runs on worker threads with --in_flight > 1, sys.stdout is process-global)
        with torch.no_grad(), autocast():
            samples = solver.sample(
                noise, steps=args.rescale_timesteps, t_start=1.0, t_end=1 / 1000, order=2, skip_type="time_uniform",
                method="adaptive" if args.adaptive else "multistep")                              # :241-249
            lat = (samples * d_std + d_mean).reshape(T, samples.shape[2], samples.shape[3])        # :250-253
            pred_delta = vae.decode(lat, padded).float()                                           # :256-259
        if probe is not None:
            probe[i] = (samples.detach().clone(), pred_delta.detach().clone(), fps4096.detach().clone(), fps512.detach().clone(), torch.stack(trace))
        # renderer input: the canonical Gaussians as a GaussianModel (the TRELLIS stage hands one over; here rebuilt from the (P, 14) tensor)
        from gvfdiffusion_amd.representations.gaussian import Gaussian
        gm = Gaussian(sh_degree=0, aabb=[-0.5, -0.5, -0.5, 1.0, 1.0, 1.0], mininum_kernel_size=synthetic.KERNEL_3D, scaling_bias=synthetic.SCALING_BIAS,
                      opacity_bias=synthetic.OPACITY_BIAS, scaling_activation="softplus", device=dev)
        gm.from_xyz(gs[:, 0:3]); gm.from_features(gs[:, 3:6].reshape(-1, 1, 3).contiguous()); gm.from_opacity(gs[:, 6:7].clamp(1e-4, 1 - 1e-4))
        gm.from_scaling(gs[:, 7:10]); gm.from_rotation(gs[:, 10:14])
        frames = torch.cat([f for _, f in render_sample_frames(rends[slot], gm, pred_delta[0], K, extrinsics=cams, n_valid=int(valid_idx[0]),
                                                               chunk_frames=96, streams=1 if n_fl > 1 else 2)])      # :261-272
        torch.cuda.current_stream().synchronize()
        stats.append((i, nfe["n"], time.perf_counter() - t0))
        return frames

    chain.models = copies              # (tests look at the DiT copies: did the solver's announced grid reach the modulation table?)
    return chain, stats, n_fl


def main(argv=None):
    args = create_argparser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("inference_dpm_latent.py needs an MI355X (the package has no CPU path)")
    from gvfdiffusion_amd import distributed as D
    from gvfdiffusion_amd.utils import seed_everything
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    rank, world = D.init_from_env(dev)
    seed_everything(args.seed + rank)
    chain, stats, n_fl = build_chain(args, dev)
    main.last_chain = chain

    t0 = time.perf_counter()
    frames, mine = D.sample_decode_render_sharded(chain, args.num_samples, device=dev, in_flight=n_fl)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if args.save_png:
        from PIL import Image
        out_dir = os.path.join(args.exp_name, "inference_images")
        os.makedirs(out_dir, exist_ok=True)
        host = frames[mine].permute(0, 1, 3, 4, 2).cpu().numpy() if len(mine) else []
        for j, i in enumerate(mine):
            for k in range(host.shape[1]):
                t, c = divmod(k, args.views)
                Image.fromarray(host[j, k]).save(os.path.join(out_dir, f"rank_{rank:02d}_render_{i:06d}_cam_{c:03d}_timesteps_{t:02d}.png"))   # :297
    if rank == 0:
        print(json.dumps({"samples": args.num_samples, "ranks": world, "frames": list(frames.shape), "wall_s": round(wall, 3),
                          "dtype": "fp16" if args.use_fp16 else "module default", "sampler": "adaptive" if args.adaptive else f"multistep x{args.rescale_timesteps}",
                          "rank0_per_sample": [{"sample": i, "nfe": n, "s": round(s, 3)} for i, n, s in stats]}))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return frames


if __name__ == "__main__":
    main()
