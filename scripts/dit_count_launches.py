"""Kernel launches of ONE DiT forward (B = 1, T = 24, configs/diffusion.yml), counted by the profiler's own activity records: run under
`rocprofv3 --kernel-trace --stats` or standalone (torch.profiler).  Prints launches per forward by kernel name."""
import collections, json, os, re, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gvfdiffusion_amd import synthetic
from gvfdiffusion_amd.model.dit import DiT

dev = torch.device("cuda:0")
man = json.load(open(os.path.join(ROOT, "tests", "golden", "dit_manifest.json")))
net = DiT(**man["config"])
net.load_state_dict(synthetic.dit_state_dict(man["state_dict"], seed=0), strict=True)
net = net.to(dev).eval()
i = {k: v.to(dev) for k, v in synthetic.dit_inputs(B=1, T=24, seed=1).items()}
kw = dict(cond_images=i["cond_images"], static_latent=i["static_latent"], deformation_position_xyz=i["deformation_position_xyz"])
net(i["x"], i["t"], **kw)                      # builds the weight / condition caches
torch.cuda.synchronize()
n = 5
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(n):
        net(i["x"], i["t"] * 0.9, **kw)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        nm = e.name.replace("(anonymous namespace)::", "")
        m = re.search(r"(\w+)(<[^(]*>)?\(", nm)
        cnt[(m.group(1) + (m.group(2) or ""))[:60] if m else nm[:60]] += 1
tot = sum(cnt.values())
for k, v in cnt.most_common():
    print(f"{v / n:7.1f}  {k}")
print(f"{tot / n:.1f} device activities (kernels + copies) per forward")
