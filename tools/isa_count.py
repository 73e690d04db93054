#!/usr/bin/env python
"""Static instruction mix of the gfx950 kernels of one HIP source, from the compiler's own assembly: which kernels are VALU-heavy, how many packed / LDS /
memory / scalar instructions a change adds or removes, registers and occupancy -- the question a VALU-bound kernel (medianBlur 5 x 5, filter2D 5 x 5, the
lean warp) asks before it asks the GPU.  No GPU needed:

    python tools/isa_count.py opencv_amd/csrc/median.hip k_median_roll          # kernels whose mangled name contains the pattern

Counts are static (a loop body counts once), so compare like with like: the same kernel before and after a change."""
import re
import subprocess
import sys
import tempfile
from collections import Counter

HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only"]


def kernels(asm, pat):
    lines = asm.split("\n")
    out, i = {}, 0
    while i < len(lines):
        m = re.match(r"^(_Z\S*%s\S*):" % re.escape(pat), lines[i])
        if m:
            ins, j = [], i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                l = lines[j].strip()
                if l and not l.startswith((".", ";", "//")) and not l.endswith(":"):
                    ins.append(l.split()[0])
                j += 1
            out[m.group(1)] = Counter(ins)
            i = j
        i += 1
    return out


def resources(asm):
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        body = m.group(2)
        v = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        a = re.search(r"\.amdhsa_accum_offset (\d+)", body)
        l = re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body)
        s = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        res[m.group(1)] = dict(vgpr_agpr=int(v.group(1)) if v else None, arch_vgpr=int(a.group(1)) if a else None, lds=int(l.group(1)) if l else 0, scratch=int(s.group(1)) if s else 0)
    return res


def main():
    src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    inc = ["-I" + src.rsplit("/", 1)[0], "-Iinclude"] if "/" in src else ["-Iinclude"]
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.check_call([HIPCC] + FLAGS + inc + ["-S", src, "-o", f.name], stderr=subprocess.DEVNULL)
        asm = open(f.name).read()
    res = resources(asm)
    for name, c in kernels(asm, pat).items():
        total = sum(c.values())
        valu = sum(n for k, n in c.items() if k.startswith("v_"))
        pk = sum(n for k, n in c.items() if k.startswith("v_pk_"))
        lds = sum(n for k, n in c.items() if k.startswith("ds_"))
        vmem = sum(n for k, n in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))
        salu = sum(n for k, n in c.items() if k.startswith("s_"))
        mfma = sum(n for k, n in c.items() if "mfma" in k)
        r = res.get(name, {})
        print("%s\n    total %d  VALU %d (packed %d, MFMA %d)  LDS %d  VMEM %d  SALU %d   registers %s (arch VGPRs %s)  LDS bytes %s  scratch %s\n    top: %s"
              % (name[:110], total, valu, pk, mfma, lds, vmem, salu, r.get("vgpr_agpr"), r.get("arch_vgpr"), r.get("lds"), r.get("scratch"),
                 ", ".join("%s %d" % kv for kv in c.most_common(8))))


if __name__ == "__main__":
    main()
