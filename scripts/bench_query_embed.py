import sys, torch, time
sys.path.insert(0, '/root/repo')
from gvfdiffusion_amd.ops import vae_ops
P, C = 262144, 768
q = torch.randn(P, 14, device='cuda'); q[:, :3] = torch.rand(P, 3, device='cuda') - 0.5
w, b = torch.randn(C, 14, device='cuda') * 0.3, torch.randn(C, device='cuda') * 0.1
E = C // 6
omega = (1.0 / 10000 ** (torch.arange(E, dtype=torch.float64) / (E / 2.0))).float().cuda()
for _ in range(3): y = vae_ops.vae_query_embed_bf16(q, w, b, omega)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): y = vae_ops.vae_query_embed_bf16(q, w, b, omega)
torch.cuda.synchronize(); print('query_embed %.3f ms' % ((time.perf_counter() - t0) / 10 * 1e3))
