"""Experiment: back-to-back rasteriser steps on ONE stream vs round-robin over N streams with N workspaces / output buffers (consecutive 4D
samples are independent: a render loop can keep several in flight).  python scripts/rast_two_streams.py [n_streams] [steps]"""
import os, sys, time, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench

dev = torch.device("cuda:0")
n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
works = [bench.RasterWorkload(dev, 262144, 800, 24, 2, seed=0)]
for i in range(1, n_streams):
    w = bench.RasterWorkload.__new__(bench.RasterWorkload)
    w.__dict__.update(works[0].__dict__)
    w.color = torch.empty_like(works[0].color); w.nr = torch.zeros_like(works[0].nr)
    w.ws = torch.empty(w.ws_bytes + 256, dtype=torch.uint8, device=dev); w.ws_base = (w.ws.data_ptr() + 255) // 256 * 256
    works.append(w)
prio = [int(p) for p in os.environ.get("GVF_PRIO", "").split(",") if p] or [0] * n_streams
streams = [torch.cuda.Stream(device=dev, priority=prio[i % len(prio)]) for i in range(n_streams)]


def run(ns):
    for i in range(4):
        with torch.cuda.stream(streams[i % ns]):
            works[i % ns].step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % ns]):
            works[i % ns].step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for ns in ([1] + list(range(2, n_streams + 1))) * 2:
    ms = run(ns)
    print(f"{ns} stream(s): {ms:.4f} ms per step, {24 / ms * 1e3:.0f} frames/s")
assert all(torch.equal(w.color, works[0].color) for w in works[1:])
