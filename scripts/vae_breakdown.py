"""Time the motion-VAE decode (released config) at BASELINE config-4 scale: 24 frames, 262144 static Gaussians."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gvfdiffusion_amd.model.autoencoder import GSKLTemporalVariationalAutoEncoder

P = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
cfg = dict(depth=12, dim=768, queries_dim=768, output_dim=14, num_inputs=8192, num_latents=512, latent_dim=16, heads=12,
           dim_head=-1, num_timesteps=24, chunk_size=8192)
torch.manual_seed(0)
m = GSKLTemporalVariationalAutoEncoder(**cfg)
with torch.no_grad():
    for p in m.parameters():
        if p.dim() == 2:
            p.copy_(torch.randn_like(p) / p.shape[1] ** 0.5)
m = m.cuda()
x = torch.randn(24, 512, 16, device="cuda")
q = torch.randn(1, P, 14, device="cuda")
q[..., :3] = torch.rand(1, P, 3, device="cuda") - 0.5
for _ in range(2):
    y = m.decode(x, q)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    h = m.decode_latents(x)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(3):
    y = m.decode(x, q)
torch.cuda.synchronize()
t2 = time.perf_counter()
lat, full = (t1 - t0) / 3 * 1e3, (t2 - t1) / 3 * 1e3
flops = 4.0 * 24 * P * 512 * 768
print(f"P={P}: latent blocks {lat:.2f} ms, full decode {full:.2f} ms; query cross-attention {flops / 1e12:.2f} TFLOP "
      f"-> {(flops / 1e12) / ((full - lat) / 1e3):.0f} TFLOP/s incl. embed/to_q/fold")
