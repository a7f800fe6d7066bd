"""Is the denoise step clock / power limited under sustained load?  ("batch buys nothing" -- the kernels of a batched forward
take 6 % less per sample in a kernel trace, profiles/r06_dit_batch_scaling.txt, the end-to-end step does not.)  Samples the shader clock and the
socket power (amdgpu sysfs / rocm-smi) every ~50 ms while the 32-step sampler runs at B = 1 and at B = 8 for a few seconds each, and prints the
per-sample-forward time beside them.   usage: python scripts/dit_clock_probe.py [seconds per leg]"""
import glob, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


def sysfs_probe():
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        sclk = os.path.join(card, "pp_dpm_sclk")
        if os.path.exists(sclk):
            pw = glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_input"))
            return sclk, (pw[0] if pw else None)
    return None, None


SCLK, POWER = sysfs_probe()


def read_once():
    mhz = watts = None
    if SCLK:
        try:
            for ln in open(SCLK):
                if "*" in ln:
                    mhz = float(re.search(r"(\d+)\s*Mhz", ln, re.I).group(1))
            if POWER:
                watts = float(open(POWER).read()) / 1e6
        except Exception:      # noqa: BLE001
            pass
    else:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            m = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
            mhz = float(m.group(1)) if m else None
            m = re.search(r"Power \(W\):\s*([\d.]+)", out) or re.search(r"Socket Power.*?:\s*([\d.]+)", out)
            watts = float(m.group(1)) if m else None
        except Exception:      # noqa: BLE001
            pass
    return mhz, watts


def leg(name, fn, per_call_samples_nfe):
    stop, rows = threading.Event(), []

    def poll():
        while not stop.is_set():
            rows.append(read_once())
            time.sleep(0.05)
    th = threading.Thread(target=poll); th.start()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < SECS:
        fn(); torch.cuda.synchronize(); n += 1
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    mhz = [r[0] for r in rows if r[0]]; w = [r[1] for r in rows if r[1]]
    per = dt / (n * per_call_samples_nfe) * 1e3 if per_call_samples_nfe else float("nan")
    print(f"{name:34s} {n:4d} calls in {dt:5.2f} s  ms per sample-forward {per:6.3f}   sclk MHz mean {sum(mhz) / max(len(mhz), 1):7.1f} min {min(mhz) if mhz else 0:6.0f} "
          f"max {max(mhz) if mhz else 0:6.0f}   power W mean {sum(w) / max(len(w), 1):6.1f} max {max(w) if w else 0:6.1f}   ({len(rows)} polls)")


dev = torch.device("cuda:0")
print("clock source:", SCLK or "rocm-smi", "| power source:", POWER or "rocm-smi")
leg("idle", lambda: time.sleep(0.2), 0)
w1 = bench.DiTWorkload(dev)
w1.sample(steps=4); w1.sample(steps=32)
leg("B = 1, 32-step sampling", lambda: w1.sample(steps=32), 32)
del w1; torch.cuda.empty_cache()
w8 = bench.DiTWorkload(dev, input_seed=list(range(1, 9)))
w8.sample(steps=4); w8.sample(steps=32)
leg("B = 8 in one forward, 32 steps", lambda: w8.sample(steps=32), 32 * 8)
leg("idle again", lambda: time.sleep(0.2), 0)
