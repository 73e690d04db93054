"""k_sepmx (opencv_amd/csrc/sepmx.hip) keeps two steps of source rows in flight as asynchronous global -> LDS loads and waits for them with COUNTED s_waitcnt vmcnt(n).  A
spilled register is reloaded through vector memory -- in order behind every row piece in flight -- so ONE spill in the walk stalls every step for a whole HBM latency
(measured: 8UC3 19 taps 12.2 -> 18.4 us per 4K frame, profiles/r06_sepmx.txt).  The compiler decides that, so the build is checked: every instantiation of the kernel
in the built object must report no spilled registers and no scratch.  CPU test: reads the code object's metadata, no GPU."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def test_no_instantiation_of_k_sepmx_spills():
    obj = os.path.join(ROOT, "opencv_amd", "csrc", "build", "sepmx.o")
    if not os.path.exists(obj) or not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")):
        pytest.skip("no built object / no llvm tools here")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "sepmx.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    kernels = [b for b in notes.split("- .agpr_count:")[1:] if "k_sepmx" in b]
    assert len(kernels) >= 32, len(kernels)
    for b in kernels:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1)) == 0, name
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1)) == 0, name
