#!/usr/bin/env python
"""VGPR / occupancy / scratch / LDS of every kernel in the named csrc files, from hipcc -Rpass-analysis=kernel-resource-usage (no GPU needed):
   python tools/kernel_resources.py corner seproll ...      (a kernel sitting one register over an occupancy step is the cheapest thing to find)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "opencv_amd", "csrc")


def demangle(names):
    try:
        return subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    except OSError:
        return names


def main():
    for f in sys.argv[1:] or ["corner"]:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden", "-ffp-contract=off",
                            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CS, f + ".hip"), "-o", "/dev/null"], capture_output=True, text=True)
        rows, cur, d = [], None, {}
        for l in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", l)
            if m:
                cur, d = m.group(1), {}
            for k in ("VGPRs", "AGPRs", "Occupancy [waves/SIMD]", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]", "TotalSGPRs"):
                m2 = re.search(re.escape(k) + r": (\d+)", l)
                if m2 and cur:
                    d[k] = int(m2.group(1))
            if "LDS Size" in l and cur:
                rows.append((cur, d)); cur = None
        names = demangle([n for n, _ in rows])
        print("==", f)
        for (n, d), dn in zip(rows, names):
            print(f"{d.get('VGPRs', 0):4d} vgpr {d.get('AGPRs', 0):3d} agpr {d.get('TotalSGPRs', 0):3d} sgpr  occ {d.get('Occupancy [waves/SIMD]')}  scratch {d.get('ScratchSize [bytes/lane]')}"
                  f"  lds {d.get('LDS Size [bytes/block]', 0):6d}  {dn[:150]}")


if __name__ == "__main__":
    main()
