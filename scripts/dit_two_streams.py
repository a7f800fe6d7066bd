"""Experiment: N independent samples denoised concurrently (one Python thread + HIP stream + DiT instance each, same weights) against one at
a time.  python scripts/dit_two_streams.py [n] [nfe]"""
import os, sys, time, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nfe = int(sys.argv[2]) if len(sys.argv) > 2 else 32
works = [bench.DiTWorkload(dev, input_seed=5 + i) for i in range(n)]
streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
outs = [None] * n


def run_one(i, steps):
    with torch.cuda.stream(streams[i]):
        outs[i] = works[i].sample(steps=steps)
    streams[i].synchronize()


def run(k, steps):
    ths = [threading.Thread(target=run_one, args=(i, steps)) for i in range(k)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for i in range(n):
    run_one(i, 4)
ref = []
for i in range(n):
    run_one(i, nfe); ref.append(outs[i].clone())
for k in [1, n, 1, n]:
    dt = run(k, nfe)
    print(f"{k} sample(s) in flight: {dt / nfe * 1e3:.3f} ms per solver step, {k * nfe / dt:.1f} denoise steps/s aggregate")
for i in range(n):
    assert torch.equal(outs[i], ref[i]), "concurrent and serial sampling must give the same latents"
