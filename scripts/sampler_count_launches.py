"""Device activities per DPM-Solver step OUTSIDE the network (the network is replaced by a stub that returns a fixed tensor)."""
import collections, os, re, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gvfdiffusion_amd.model.dpmsolver import NoiseScheduleVP, model_wrapper, DPM_Solver
from gvfdiffusion_amd.model.gaussian_diffusion import create_gaussian_diffusion

dev = torch.device("cuda:0")
ns = NoiseScheduleVP("discrete", betas=torch.from_numpy(create_gaussian_diffusion(noise_schedule="cosine", predict_type="v").betas))
x = torch.randn((1, 24, 512, 16), device=dev)
y = torch.randn_like(x)
guided = len(sys.argv) > 1
kw = dict(guidance_type="classifier-free", guidance_scale=3.0, guidance_scale2=1.5, condition={"c": torch.zeros(1, device=dev)},
          unconditional_condition={"c": torch.zeros(1, device=dev)}) if guided else {}
fn = model_wrapper(lambda x_, t_, **k: y.expand(x_.shape[0], -1, -1, -1) * 1.0, ns, model_type="v", model_kwargs={}, **kw)
solver = DPM_Solver(fn, ns, algorithm_type="dpmsolver++")
solver.sample(x, steps=4, t_start=1.0, t_end=1 / 1000, order=2, skip_type="time_uniform", method="multistep")
torch.cuda.synchronize()
n = 32
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    solver.sample(x, steps=n, t_start=1.0, t_end=1 / 1000, order=2, skip_type="time_uniform", method="multistep")
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        m = re.search(r"(\w+)(<[^(]*>)?\(", e.name)
        cnt[(m.group(1) + (m.group(2) or ""))[:90] if m else e.name[:90]] += 1
for k, v in cnt.most_common():
    print(f"{v / n:7.2f}  {k}")
print(f"{sum(cnt.values()) / n:.1f} device activities per solver step (the stub's own multiply included)")
