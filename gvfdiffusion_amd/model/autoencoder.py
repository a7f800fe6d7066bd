"""Motion VAE (`GSKLTemporalVariationalAutoEncoder`) on the MI355X kernels: decode (the inference path) and encode.

Mirrors model/autoencoder.py:345-609 of the reference: same constructor keywords, same parameter tree (all 121
tensors of the released checkpoint load with ``strict=True``), same ``decode(x, queries)`` signature and
output ``(B, T, P, output_dim)``.  Only inference-time ``decode`` is on the hot path (inference_dpm_latent.py:
252-256 -> utils/inference_utils.py: pred_delta = vae.decode(latents, static_gs)); ``encode`` (encode_latent.py,
training) runs on the same kernels plus csrc/fps.hip for the farthest point sampling; its KNN interpolation is a
brute-force torch.topk (pytorch3d.ops.knn_points upstream).

What differs from the reference's op order (results agree to the bf16 tolerance stated in tests/test_vae_gpu.py):
  * the query embedding (gs_embedding + position_encoding + PreNorm LN) and the decoder to_q projection depend on
    the static Gaussians only; the reference repeats them for each of the T frames (process_chunk :557), here they
    run once per sample and the attention kernel reads the same q for every frame (inner stride 0);
  * to_out (dim -> dim) and to_outputs (dim -> output_dim) have nothing between them (decoder_ff is None in the
    released config), so they are folded into one (output_dim x dim) matrix at weight-preparation time;
  * to_q / to_kv of the latent self-attention run as one N = 3*dim GEMM; K and V^T of the decoder cross-attention
    are laid out head-major once per decode (as the DiT's condition cache).
Precision placement = the reference under autocast(bf16): bf16 GEMM/attention operands, fp32 accumulate,
fp32 residual stream, LayerNorm and softmax.
"""

import numpy as np
import os
import torch
import torch.nn as nn

from .. import _lib
from ..ops import dit_ops, vae_ops
from ..ops import precision


class GEGLU(nn.Module):
    def forward(self, x):  # parameter-less placeholder keeping `net.2` at index 2 (model/autoencoder.py:90-93)
        raise RuntimeError("GEGLU runs inside gvf_geglu_bf16; call GSKLTemporalVariationalAutoEncoder.decode")


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, dim * mult * 2), GEGLU(), nn.Linear(dim * mult, dim))


class Attention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_kv = nn.Linear(context_dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, query_dim)


class PreNorm(nn.Module):
    """Holds `fn`; its LayerNorms (eps 1e-6) have no parameters (model/autoencoder.py:73-88)."""

    def __init__(self, dim, fn, context_dim=None):
        super().__init__()
        self.fn = fn


class PointEmbed(nn.Module):
    def __init__(self, hidden_dim=48):
        super().__init__()
        assert hidden_dim % 6 == 0
        self.embedding_dim = hidden_dim // 3 // 2
        omega = np.arange(self.embedding_dim, dtype=np.float64)
        omega /= self.embedding_dim / 2.0
        self.register_buffer("omega", torch.from_numpy(1.0 / 10000 ** omega))


class DiagonalGaussianDistribution:
    """model/autoencoder.py:304-342 (the members encode()'s callers use)."""

    def __init__(self, mean, logvar, deterministic=False):
        self.mean = mean
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape, device=self.mean.device)

    def kl(self):
        if self.deterministic:
            return torch.zeros(1)
        return 0.5 * torch.mean(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=[1, 2])

    def mode(self):
        return self.mean


class GSKLTemporalVariationalAutoEncoder(nn.Module):
    def __init__(self, *, depth=24, dim=512, queries_dim=512, input_dim=3, gs_dim=14, output_dim=10, num_inputs=8192,
                 num_latents=1024, latent_dim=128, heads=8, dim_head=-1, weight_tie_layers=False, decoder_ff=False,
                 enable_flash_attn=False, num_timesteps=24, chunk_size=8192, knn_k=8, beta=7.0):
        super().__init__()
        if dim_head == -1:
            dim_head = dim // heads
        if decoder_ff or weight_tie_layers:
            raise NotImplementedError("decoder_ff / weight_tie_layers are off in every released config")
        if dim_head not in (32, 64):
            raise ValueError(f"gvf_attn_fwd_bf16 supports head_dim 32 or 64, got {dim_head}")
        if queries_dim != dim or dim % 64 != 0 or dim % 6 != 0:
            raise ValueError("dim must equal queries_dim and be a multiple of 64 (GEMM K) and of 6 (PointEmbed)")
        self.depth, self.dim, self.heads, self.dim_head = depth, dim, heads, dim_head
        self.num_inputs, self.num_latents, self.num_timesteps = num_inputs, num_latents, num_timesteps
        self.knn_k, self.beta, self.chunk_size, self.output_dim, self.gs_dim = knn_k, beta, chunk_size, output_dim, gs_dim

        def attn(qd, cd=None):
            return Attention(qd, cd, heads=heads, dim_head=dim_head)

        # encoder half: parameters only (so that released checkpoints load strictly)
        self.cross_attend_blocks = nn.ModuleList([PreNorm(dim, attn(dim, dim), context_dim=dim), PreNorm(dim, FeedForward(dim))])
        self.input_embedding = nn.Sequential(nn.Linear(input_dim, dim), nn.LayerNorm(dim, elementwise_affine=False))
        self.gs_embedding = nn.Sequential(nn.Linear(gs_dim, dim), nn.LayerNorm(dim, elementwise_affine=False))
        self.position_encoding = nn.Sequential(PointEmbed(hidden_dim=dim), nn.LayerNorm(dim, elementwise_affine=False))
        self.layers = nn.ModuleList([nn.ModuleList([PreNorm(dim, attn(dim)), PreNorm(dim, FeedForward(dim))]) for _ in range(depth)])
        self.decoder_cross_attn = PreNorm(queries_dim, attn(queries_dim, dim), context_dim=dim)
        self.decoder_ff = None
        self.to_outputs = nn.Linear(queries_dim, output_dim)
        self.proj = nn.Linear(latent_dim, dim)
        self.mean_fc = nn.Linear(dim, latent_dim)
        self.logvar_fc = nn.Linear(dim, latent_dim)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        nn.init.zeros_(self.to_outputs.weight)       # zero_module(to_outputs), model/autoencoder.py:434
        nn.init.zeros_(self.to_outputs.bias)
        self._wcache = None
        self.compute_dtype = None                    # None: ops/precision.py decides per call (GVF_DIT_DTYPE, an autocast region -- the
                                                     # reference decodes under accelerate's fp16 autocast, inference_dpm_latent.py:256 --, else bf16)
        self.max_chunk_rows = (1 << 20) + (1 << 16)  # hard cap on the rows (B*T*Pc) of the 16-bit attention-output chunk: 1.6 GiB at dim 768 (released config: 6 chunks of <= 45056 Gaussians x 24 frames)

    def set_compute_dtype(self, dtype):
        """torch.float16 / torch.bfloat16 (or "fp16" / "bf16"): the matrix pipe's operand type; None = ops/precision.py's rule."""
        self.compute_dtype = precision.parse(dtype)
        return self

    def _lp(self):
        return precision.resolve(self.compute_dtype, (), torch.bfloat16)

    # ---- encode (encode_latent.py / training; not on the inference path) ---------------------------------------------
    @staticmethod
    @torch.no_grad()
    def compute_delta_interp(static_gs, micro_static_pc, micro_moving_pc, knn_k=8, beta=7.0, adaptive_radius=True):
        """KNN-interpolated motion of the sampled Gaussians (model/autoencoder.py:449-500; pytorch3d.ops.knn_points is a
        brute-force K-nearest search here: squared distances ascending).  static_gs (B,L,3), micro_static_pc (B,N,3),
        micro_moving_pc (B,T,N,3) -> (B,T,L,3)."""
        d2 = ((static_gs[:, :, None, :] - micro_static_pc[:, None, :, :]) ** 2).sum(-1)             # (B, L, N)
        knn_dists, knn_idx = torch.topk(d2, knn_k, dim=-1, largest=False, sorted=True)
        radii = knn_dists.mean(dim=-1).sqrt() + 1e-6
        if adaptive_radius:
            w = torch.exp(-beta * knn_dists / radii[..., None] ** 2) * (knn_dists <= radii[..., None] ** 2).float()
        else:
            w = torch.exp(-beta * knn_dists)
        w = w / (w.sum(dim=-1, keepdim=True) + 1e-8)
        B, L, K = knn_idx.shape
        T = micro_moving_pc.shape[1]
        idx = knn_idx.reshape(B, 1, L * K, 1).expand(B, T, L * K, 3)
        nb = torch.gather(micro_moving_pc, 2, idx).reshape(B, T, L, K, 3)                            # neighbour positions per frame
        nb0 = torch.gather(micro_static_pc, 1, knn_idx.reshape(B, L * K, 1).expand(B, L * K, 3)).reshape(B, 1, L, K, 3)
        return ((nb - nb0) * w[:, None, :, :, None]).sum(dim=3)

    @torch.no_grad()
    def encode(self, static_pc, delta_pc, static_gs_list, random_start: bool = True, sample_posterior: bool = True):
        """static_pc (B,N,3), delta_pc (B,T,N,3), static_gs_list [ (N_gs,14) ] -> (kl, x, posterior, sampled_static_gs)
        (model/autoencoder.py:502-550).  random_start=False makes the farthest point sampling start at each sample's
        first Gaussian (upstream's default is a random start)."""
        from ..utils.points import sample_gs
        _lib.require_cuda(static_pc, delta_pc)
        W = self._weights()
        B, N, _ = static_pc.shape
        T = delta_pc.shape[1]
        C, H, d, L, dev = self.dim, self.heads, self.dim_head, self.num_latents, static_pc.device
        bf16 = W["lp"]                                   # the 16-bit operand type of this call (bf16 or fp16)
        sampled_static_gs = sample_gs(static_gs_list, L, random_start=random_start)                  # (B, L, 14)
        input_static_gs = sampled_static_gs[..., :3].float().contiguous()
        static_pc, delta_pc = static_pc.float(), delta_pc.float()
        moving_pc = delta_pc + static_pc[:, None]
        est = self.compute_delta_interp(input_static_gs, static_pc, moving_pc, knn_k=self.knn_k, beta=self.beta)   # (B,T,L,3)
        # embeddings: rows [xyz | delta]; the Linear sees only the delta (zero weights on xyz), PointEmbed only the xyz
        qrows = torch.cat([input_static_gs[:, None].expand(B, T, L, 3), est], dim=-1).reshape(B * T * L, 6).contiguous()
        crows = torch.cat([static_pc[:, None].expand(B, T, N, 3), delta_pc], dim=-1).reshape(B * T * N, 6).contiguous()
        xn, x = vae_ops.vae_embed_bf16_f32(qrows, W["in_w6"], W["in_b"], W["omega"], dtype=bf16)
        cn, _ = vae_ops.vae_embed_bf16_f32(crows, W["in_w6"], W["in_b"], W["omega"], want_embed=False, dtype=bf16)
        M, Mc = B * T * L, B * T * N
        e = W["enc"]
        q = torch.empty((M, C), dtype=bf16, device=dev)
        dit_ops.gemm_bf16(xn, e["q"], None, q, dit_ops.EPI_STORE_BF16)
        kv = torch.empty((Mc, 2 * C), dtype=bf16, device=dev)
        dit_ops.gemm_bf16(cn, e["kv"], None, kv, dit_ops.EPI_STORE_BF16)
        del cn
        ao = torch.empty((M, C), dtype=bf16, device=dev)
        skv = (N * 2 * C, 0, 2 * C)
        dit_ops.attention_bf16(q, kv, kv[:, C:], ao, B * T, 1, L, N, H, (L * C, 0, C), skv, skv, (L * C, 0, C), head_dim=d)
        dit_ops.gemm_bf16(ao, *e["out"], x, dit_ops.EPI_RESID_F32)
        hb = torch.empty((M, C), dtype=bf16, device=dev)
        dit_ops.layernorm_modulate_bf16(x, hb, 1e-6)
        act = torch.empty((M, 4 * C), dtype=bf16, device=dev)
        dit_ops.gemm_bf16(hb, *e["fc1g"], act, dit_ops.EPI_GEGLU_16)            # fc1 + GEGLU in one launch (rows interleaved at pack time)
        dit_ops.gemm_bf16(act, *e["fc2"], x, dit_ops.EPI_RESID_F32)
        xb = dit_ops.cast_pad(x, C, dtype=bf16)
        Dl = self.mean_fc.out_features
        mean = torch.empty((M, Dl), dtype=torch.float32, device=dev)
        logvar = torch.empty((M, Dl), dtype=torch.float32, device=dev)
        dit_ops.gemm_bf16(xb, *e["mean"], mean, dit_ops.EPI_STORE_F32)
        dit_ops.gemm_bf16(xb, *e["logvar"], logvar, dit_ops.EPI_STORE_F32)
        posterior = DiagonalGaussianDistribution(mean.view(B * T, L, Dl), logvar.view(B * T, L, Dl))
        z = posterior.sample() if sample_posterior else posterior.mode()
        return posterior.kl(), z, posterior, sampled_static_gs

    def pad_static_gs(self, static_gs):
        """model/autoencoder.py:611-619."""
        from ..utils.points import pad_static_gs
        return pad_static_gs(static_gs)

    def forward(self, static_gs, static_pc, delta_pc, **kw):
        """encode -> decode over the padded static Gaussians (model/autoencoder.py:621-627; same argument order)."""
        kl, x, posterior, _ = self.encode(static_pc, delta_pc, static_gs, **kw)
        padded, _ = self.pad_static_gs(static_gs)
        return {"logits": self.decode(x, padded).squeeze(-1), "kl": kl, "posterior": posterior}

    # ---- weights -----------------------------------------------------------------------------------------
    def _param_version(self):
        return tuple((p._version, p.data_ptr()) for p in self.parameters())

    def _weights(self):
        lp = self._lp()
        ver = (self._param_version(), lp)
        if self._wcache is not None and self._wcache["ver"] == ver:
            return self._wcache

        def bf(w):
            w = w.detach().float().contiguous()
            return dit_ops.cast_pad(w, dit_ops.pad64(w.shape[1]), dtype=lp)

        def fb(b):
            return None if b is None else b.detach().float().contiguous()

        def glu(lin):                   # GEGLU projection: rows interleaved for the GEMM's EPI_GEGLU_16 epilogue
            wi, bi = dit_ops.geglu_interleave(lin.weight.detach().float(), None if lin.bias is None else lin.bias.detach().float())
            return bf(wi), fb(bi)

        W = {"ver": ver, "lp": lp, "proj": (bf(self.proj.weight), fb(self.proj.bias)), "layers": []}
        for a, f in self.layers:
            W["layers"].append(dict(
                qkv=bf(torch.cat([a.fn.to_q.weight, a.fn.to_kv.weight], 0)),
                out=(bf(a.fn.to_out.weight), fb(a.fn.to_out.bias)),
                fc1g=glu(f.fn.net[0]),
                fc2=(bf(f.fn.net[2].weight), fb(f.fn.net[2].bias))))
        d = self.decoder_cross_attn.fn
        W["dec_q"], W["dec_kv"] = bf(d.to_q.weight), bf(d.to_kv.weight)
        # (on the HOST: 14 x 768 x 768 in double.  A torch GEMM on the device creates a hipBLASLt handle on the calling thread's first use, which
        # touches the legacy stream -- illegal while another in-flight slot captures its hipGraph: hipBLASLt answers error 906 with exit(1).)
        dev_w = d.to_out.weight.device
        wo, wy = d.to_out.weight.detach().double().cpu(), self.to_outputs.weight.detach().double().cpu()
        W["fold"] = (bf((wy @ wo).float().to(dev_w)),
                     (wy @ d.to_out.bias.detach().double().cpu() + self.to_outputs.bias.detach().double().cpu()).float().contiguous().to(dev_w))
        # encoder half
        c = self.cross_attend_blocks
        W["enc"] = dict(q=bf(c[0].fn.to_q.weight), kv=bf(c[0].fn.to_kv.weight), out=(bf(c[0].fn.to_out.weight), fb(c[0].fn.to_out.bias)),
                        fc1g=glu(c[1].fn.net[0]), fc2=(bf(c[1].fn.net[2].weight), fb(c[1].fn.net[2].bias)),
                        mean=(bf(self.mean_fc.weight), fb(self.mean_fc.bias)), logvar=(bf(self.logvar_fc.weight), fb(self.logvar_fc.bias)))
        wi = self.input_embedding[0].weight.detach().float()                      # (dim, input_dim = 3): acts on the delta columns
        W["in_w6"] = torch.cat([torch.zeros_like(wi), wi], dim=1).contiguous()     # rows are [xyz | delta]
        W["in_b"] = self.input_embedding[0].bias.detach().float().contiguous()
        W["gs_w"] = self.gs_embedding[0].weight.detach().float().contiguous()
        W["gs_b"] = self.gs_embedding[0].bias.detach().float().contiguous()
        W["omega"] = self.position_encoding[0].omega.detach().float().contiguous()
        self._wcache = W
        return W

    # ---- decode ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode_latents(self, x: torch.Tensor) -> torch.Tensor:
        """proj + `depth` latent self-attention / GEGLU blocks: (B*T, L, latent_dim) -> fp32 (B*T*L, dim)."""
        _lib.require_cuda(x)
        W = self._weights()
        BT, L, Dl = x.shape
        C, H, d, dev = self.dim, self.heads, self.dim_head, x.device
        M = BT * L
        bf16 = W["lp"]                                   # the 16-bit operand type of this call (bf16 or fp16)
        xb = dit_ops.cast_pad(x.reshape(M, Dl).float().contiguous(), dit_ops.pad64(Dl), dtype=bf16)
        h = torch.empty((M, C), dtype=torch.float32, device=dev)
        dit_ops.gemm_bf16(xb, *W["proj"], h, dit_ops.EPI_STORE_F32)
        hb = torch.empty((M, C), dtype=bf16, device=dev)
        qkv = torch.empty((M, 3 * C), dtype=bf16, device=dev)
        act = torch.empty((M, 4 * C), dtype=bf16, device=dev)
        s3 = (L * 3 * C, 0, 3 * C)
        for w in W["layers"]:
            dit_ops.layernorm_modulate_bf16(h, hb, 1e-6)
            dit_ops.gemm_bf16(hb, w["qkv"], None, qkv, dit_ops.EPI_STORE_BF16)
            dit_ops.attention_bf16(qkv, qkv[:, C:], qkv[:, 2 * C:], hb, BT, 1, L, L, H, s3, s3, s3, (L * C, 0, C), head_dim=d)
            dit_ops.gemm_bf16(hb, *w["out"], h, dit_ops.EPI_RESID_F32)
            dit_ops.layernorm_modulate_bf16(h, hb, 1e-6)
            # fc1 + GEGLU in one launch: the projection's rows are interleaved (32 value rows, their 32 gate rows) when the weights are packed, the
            # epilogue pairs them up -- bit-identical to the store epilogue + gvf_geglu, without the 6144-wide intermediate (151 MB written and re-read per block)
            dit_ops.gemm_bf16(hb, *w["fc1g"], act, dit_ops.EPI_GEGLU_16)
            dit_ops.gemm_bf16(act, *w["fc2"], h, dit_ops.EPI_RESID_F32)
        return h

    @torch.no_grad()
    def decode(self, x: torch.Tensor, queries: torch.Tensor) -> torch.Tensor:
        """x: (B*T, L, latent_dim), queries: (B, P, gs_dim) -> (B, T, P, output_dim)  (model/autoencoder.py:579-609)."""
        _lib.require_cuda(x, queries)
        W = self._weights()
        B, P = queries.shape[:2]
        T, C, H, d, dev = self.num_timesteps, self.dim, self.heads, self.dim_head, x.device
        BT, L = x.shape[:2]
        if BT != B * T:
            raise ValueError(f"x has {BT} latent sets, queries imply B*T = {B}*{T}")
        bf16 = W["lp"]                                   # the 16-bit operand type of this call (bf16 or fp16)
        h = self.decode_latents(x)
        # context: PreNorm.norm_context -> to_kv, K head-major and V^T zero-padded to 64 keys, per (b, t)
        hb = torch.empty((BT * L, C), dtype=bf16, device=dev)
        dit_ops.layernorm_modulate_bf16(h, hb, 1e-6)
        kv = torch.empty((BT * L, 2 * C), dtype=bf16, device=dev)
        dit_ops.gemm_bf16(hb, W["dec_kv"], None, kv, dit_ops.EPI_STORE_BF16)
        # K / V^T of the (b, t) latent sets in the image the attention stages (K pre-scaled): csrc/attn_xt64.hip when the key set fits its
        # LDS (head_dim 64, <= 512 latents: the released config), else head-major K and zero-padded V^T for csrc/attn.hip
        tiled = d == 64 and L <= 512 and os.environ.get("GVF_VAE_TILED64", "1") != "0"
        if tiled:
            kt, vt = dit_ops.attention_pack_kv64(kv, BT, L, H, 0, C)
        else:
            kc = kv[:, :C].reshape(BT, L, H, d).permute(0, 2, 1, 3).contiguous()
            Lp = (L + 63) // 64 * 64
            vt = torch.zeros((BT, H, d, Lp), dtype=bf16, device=dev)
            vt[..., :L] = kv[:, C:].reshape(BT, L, H, d).permute(0, 2, 3, 1)
        # queries: embedding + PreNorm + to_q once per static Gaussian (shared by the T frames)
        qe = vae_ops.vae_query_embed_bf16(queries.reshape(B * P, -1).float().contiguous(), W["gs_w"], W["gs_b"], W["omega"], dtype=bf16)
        qp = torch.empty((B * P, C), dtype=bf16, device=dev)
        dit_ops.gemm_bf16(qe, W["dec_q"], None, qp, dit_ops.EPI_STORE_BF16)
        del qe
        out = torch.empty((B, T, P, self.output_dim), dtype=torch.float32, device=dev)
        # chunks of about max_chunk_rows attention-output rows, all the same size up to the attention's 2048-query workgroups (the released
        # config: 5 x 45056 + 36864 Gaussians; a fixed 43648 left a 256-Gaussian seventh launch pair)
        # B * T * Pc <= max_chunk_rows is a CAP (the attention-output / partials buffer it sizes): ceil on the chunk count, and the chunk rounded
        # UP to the workgroup granularity only while that keeps it under the cap, otherwise down (never below one workgroup's 2048 queries)
        n_chunks = max(1, -(-(B * T * P) // self.max_chunk_rows))
        cap = max(2048, self.max_chunk_rows // (B * T) // 2048 * 2048)
        Pc = min(P, cap, ((P + n_chunks - 1) // n_chunks + 2047) // 2048 * 2048)
        qp3 = qp.view(B, P, C)
        if tiled and self.output_dim <= 16 and os.environ.get("GVF_VAE_FOLD", "1") != "0":
            # to_out o to_outputs applied per head in the attention's epilogue: 16 fp32 partial products per (frame, Gaussian, head) instead of
            # the 768-wide 16-bit rows and the GEMM that re-read them; the heads are added in ascending order by the reduce launch
            frags = W.get("fold_frags")
            if frags is None:
                frags = W["fold_frags"] = dit_ops.attention_fold_pack(W["fold"][0], self.output_dim, H)
            part = torch.empty((B * T, H, Pc, 16), dtype=torch.float32, device=dev)
            for p0 in range(0, P, Pc):
                n = min(Pc, P - p0)
                dit_ops.attention_tiled64_fold(qp3[:, p0:], kt, vt, frags, part, B, T, n, L, H, (P * C, 0, C), T, 1)
                dit_ops.attention_fold_reduce(part, W["fold"][1], out[:, :, p0:], B * T, H, n, self.output_dim, P * self.output_dim, self.output_dim)
            return out.to(queries.dtype if queries.dtype.is_floating_point else torch.float32)
        ao = torch.empty((B, T, Pc, C), dtype=bf16, device=dev)
        yo = torch.empty((B * T * Pc, self.output_dim), dtype=torch.float32, device=dev)
        for p0 in range(0, P, Pc):
            n = min(Pc, P - p0)
            a = ao if n == Pc else ao.view(-1)[:B * T * n * C].view(B, T, n, C)
            if tiled:
                dit_ops.attention_tiled64(qp3[:, p0:], kt, vt, a, B, T, n, L, H, (P * C, 0, C), (T * n * C, n * C, C), T, 1)
            else:
                dit_ops.attention_bf16(qp3[:, p0:], kc, vt, a, B, T, n, L, H, (P * C, 0, C), (T * H * L * d, H * L * d, d, L * d),
                                       (T * H * d * Lp, H * d * Lp, Lp, d * Lp), (T * n * C, n * C, C), v_transposed=True, head_dim=d)
            y = yo[:B * T * n]
            dit_ops.gemm_bf16(a.view(B * T * n, C), *W["fold"], y, dit_ops.EPI_STORE_F32)
            out[:, :, p0:p0 + n] = y.view(B, T, n, self.output_dim)
        return out.to(queries.dtype if queries.dtype.is_floating_point else torch.float32)
