"""Reproducer / bisection harness for "two samples in flight + one hipGraph capture per sample returns a sample that differs in its last bits"
(3 of 50 runs of inference_dpm_latent.py --in_flight 2 in round 3).  ROOT CAUSE (round 4, profiles/r04_inflight_root_cause.txt): not
the capture -- packed-fp32 VALU arithmetic (v_pk_fma_f32 & co., emitted by clang's SLP vectoriser into modulation_f32_kernel) returns wrong values
on gfx950 while another wave of the same CU issues MFMAs; one sample's DiT step shared CUs with the other sample's VAE-decode GEMMs.  With the
library built without packed fp32 (gvfdiffusion_amd/_build.py) this harness reports 0 divergent rounds, captures in flight included.

One process, models built once.  Serial references of K samples first (slot 0, graphs on), then R rounds of the same K samples with two in
flight -- every sample a NEW capture on its slot's thread while the other slot runs -- each compared with its reference at three points:
the sampler's latents, the decoded deltas, the uint8 frames.  Switches (environment) isolate the suspects:
  REPRO_EAGER=1        in-flight instances launch eagerly (the shipped rule)                         -> expected clean
  REPRO_NO_EMPTY=1     torch.cuda.empty_cache() is a no-op while a capture begins (torch.cuda.graph.__enter__ calls it)
  REPRO_NO_GC=1        gc.collect() likewise
  REPRO_NO_SYNC=1      torch.cuda.synchronize() likewise (the two device-wide syncs of a capture)
  REPRO_PAUSE=1        the OTHER slot is paused (at its next model call) while a slot captures: capture never overlaps foreign launches
  REPRO_DET=1          torch.use_deterministic_algorithms(True): rocBLAS without atomics
  REPRO_SOLOCOND=1     DiT.prepare_conditions (the hoisted fp32 library GEMMs) runs with the device to itself: every other slot is held at its
                       next model call and the device is drained before and after
  REPRO_ROUNDS, REPRO_SAMPLES, REPRO_VIEWS, REPRO_GAUSSIANS, REPRO_STEPS

    python scripts/inflight_capture_repro.py            # prints one line per divergent sample and a summary
"""
import gc
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    env = os.environ.get
    eager = env("REPRO_EAGER") == "1"
    os.environ["GVF_INFLIGHT_GRAPH"] = "0" if eager else "1"
    import inference_dpm_latent as S
    from gvfdiffusion_amd.utils.in_flight import run_in_flight
    from gvfdiffusion_amd.model import dit as dit_mod
    K, R = int(env("REPRO_SAMPLES", "4")), int(env("REPRO_ROUNDS", "25"))
    argv = ["--synthetic", "--num_samples", str(K), "--use_fp16", "--in_flight", "2", "--gaussians", env("REPRO_GAUSSIANS", "32768"),
            "--views", env("REPRO_VIEWS", "2"), "--resolution", "256"]
    if env("REPRO_STEPS"):
        argv += ["--rescale_timesteps", env("REPRO_STEPS")]
    else:
        argv += ["--adaptive"]
    args = S.create_argparser().parse_args(argv)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    probe = {}
    chain, stats, n_fl = S.build_chain(args, dev, probe=probe)

    # ---- the suspects inside torch.cuda.graph.__enter__: device-wide sync, gc, empty_cache -- switched off only while a capture begins
    real_empty, real_gc, real_sync = torch.cuda.empty_cache, gc.collect, torch.cuda.synchronize
    if env("REPRO_NO_EMPTY") == "1":
        torch.cuda.empty_cache = lambda: None
    if env("REPRO_NO_GC") == "1":
        gc.collect = lambda *a, **k: 0
    if env("REPRO_NO_SYNC") == "1":
        torch.cuda.synchronize = lambda *a, **k: None
    # ---- REPRO_PAUSE: a slot that wants to capture takes the gate exclusively; every model call of the other slot passes through it
    gate = threading.Lock()
    if env("REPRO_PAUSE") == "1":
        real_fwd = dit_mod.DiT._forward_graphed

        def gated(self, *a, **k):
            with gate:
                pass                                      # wait while a capture is in progress elsewhere
            return real_fwd(self, *a, **k)
        real_cap_lock = dit_mod._CAPTURE_LOCK

        class Both:
            def __enter__(self_):
                gate.acquire(); torch.cuda.synchronize(); real_cap_lock.acquire()
            def __exit__(self_, *e):
                real_cap_lock.release(); gate.release()
        dit_mod._CAPTURE_LOCK = Both()
        dit_mod.DiT._forward_graphed = gated

    if env("REPRO_DET") == "1":
        torch.use_deterministic_algorithms(True, warn_only=True)
    if env("REPRO_SOLOCOND") == "1":
        cond_gate = threading.Lock()
        real_prep = dit_mod.DiT.prepare_conditions
        real_fwd2 = dit_mod.DiT._forward

        def solo_prep(self, *a, **k):
            hit = getattr(self, "_ctx_cache", None)
            with cond_gate:
                torch.cuda.synchronize()
                out = real_prep(self, *a, **k)
                torch.cuda.synchronize()
            return out
        dit_mod.DiT.prepare_conditions = solo_prep

    def job(slot, i):
        f = chain(slot, i)
        x0, delta, f4096, f512, trace = probe[i]
        return x0, delta, f, f4096, f512, trace

    with torch.no_grad():
        t0 = time.time()
        ref = [tuple(t.clone() for t in job(0, i)) for i in range(K)]
        torch.cuda.synchronize()
        print(f"# serial references of {K} samples: {time.time() - t0:.1f} s; NFE per sample {[n for _, n, _ in stats]}", flush=True)
        bad_rounds, events = 0, []
        t0 = time.time()
        for r in range(R):
            res = run_in_flight([lambda slot, i=i: job(slot, i) for i in range(K)], dev, 2)
            torch.cuda.synchronize()
            bad = False
            for i in range(K):
                d = [a.shape != b.shape or not torch.equal(a, b) for a, b in zip(res[i], ref[i])]
                if any(d):
                    bad = True
                    x0, delta, f = res[i][:3]
                    ev = {"round": r, "sample": i, "latents_differ": d[0], "deltas_differ": d[1], "frames_differ": d[2], "fps4096_differ": d[3],
                          "fps512_differ": d[4], "fps4096_rows": int((res[i][3] != ref[i][3]).any(dim=-1).sum()),
                          "nfe": [int(res[i][5].shape[0]), int(ref[i][5].shape[0])],
                          "first_divergent_evaluation": int((res[i][5][:min(res[i][5].shape[0], ref[i][5].shape[0])] != ref[i][5][:min(res[i][5].shape[0], ref[i][5].shape[0])]).nonzero()[:1].sum()) if (res[i][5][:min(res[i][5].shape[0], ref[i][5].shape[0])] != ref[i][5][:min(res[i][5].shape[0], ref[i][5].shape[0])]).any() else -1,
                          "latent_max_abs": float((x0 - ref[i][0]).abs().max()), "latent_frac": float((x0 != ref[i][0]).float().mean()),
                          "delta_max_abs": float((delta - ref[i][1]).abs().max()),
                          "frame_frac": float((f != ref[i][2]).float().mean())}
                    events.append(ev)
                    print(json.dumps(ev), flush=True)
            bad_rounds += bad
        summary = {"rounds": R, "samples_per_round": K, "divergent_rounds": bad_rounds, "divergent_samples": len(events),
                   "seconds": round(time.time() - t0, 1), "switches": {k: v for k, v in os.environ.items() if k.startswith("REPRO_")}}
        print(json.dumps(summary), flush=True)
        return summary


if __name__ == "__main__":
    main()
