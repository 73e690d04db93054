"""Guards on the generated gfx950 code (CPU only: hipcc cross-compiles without a GPU).

v_ashr_pk_u8_i32: hipcc (ROCm 7.2) folds `clamp(x >> n)` of two values that are then packed into bytes 0 and 1 into this instruction and ORs further bytes into the
upper half of its result as if it were zero; on the MI355X it is not (GPU call r04f: stray bits in byte 2 of CV_8UC4 pixels out of k_warp_taps_lds).
The kernels that pack clamped shifts keep the value opaque between shift and clamp; this test fails if the instruction comes back into the files that have such code."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc")
@pytest.mark.parametrize("source", ["warp.hip"])
def test_no_ashr_pk_u8_in_the_packing_kernels(source):
    obj = os.path.join(ROOT, "opencv_amd", "csrc", "build", source.replace(".hip", ".o"))
    llvm = "/opt/rocm/lib/llvm/bin"
    src_path = os.path.join(ROOT, "opencv_amd", "csrc", source)
    if os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src_path) and os.path.exists(os.path.join(llvm, "llvm-objdump")):
        # the object __graft_entry__.build() made from this very source: unbundle its gfx950 code object and disassemble it (seconds instead of a recompile)
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
            subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
            subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
            asm = subprocess.check_output([os.path.join(llvm, "llvm-objdump"), "-d", "--mcpu=gfx950", co], text=True)
    else:
        hipcc = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")
        p = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", source, "-o", "-"],
                           cwd=os.path.join(ROOT, "opencv_amd", "csrc"), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        asm = p.stdout
    assert "s_endpgm" in asm                                          # it is the device assembly
    hits = [l.strip() for l in asm.splitlines() if "v_ashr_pk_u8_i32" in l]
    assert not hits, hits[:5]
