"""Per-kernel register / LDS / spill report of a csrc/*.hip source, compiled device-only with the flags gvfdiffusion_amd/_build.py uses
(hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).   usage: python scripts/kernel_resources.py attn_xt.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gvfdiffusion_amd import _build   # noqa: E402


def main():
    src = sys.argv[1]
    extra = sys.argv[2:]
    cmd = [_build._hipcc()] + _build.COMMON + _build.SOURCES[os.path.basename(src)] + extra + \
        ["--cuda-device-only", "-c", os.path.join(_build.CSRC, os.path.basename(src)), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        sys.exit(r.returncode)
    cur = None
    rows = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = t.split(":", 1)[1].strip()
            rows[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1)
            rows[cur][k.strip()] = v.strip()
    for name, d in rows.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        dem = re.sub(r"\(anonymous namespace\)::", "", dem)
        dem = dem.split("(")[0][-70:]
        print(f"{dem:70s} vgpr {d.get('VGPRs', '?'):>4s} agpr {d.get('AGPRs', '?'):>4s} spill {d.get('VGPRs Spill', '?'):>3s} scratch {d.get('ScratchSize [bytes/lane]', '?'):>4s} "
              f"occ {d.get('Occupancy [waves/SIMD]', '?'):>2s} lds {d.get('LDS Size [bytes/block]', '?'):>6s}")


if __name__ == "__main__":
    main()
