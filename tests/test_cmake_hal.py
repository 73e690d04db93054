"""The HAL cmake package (cmake/hal/OpenCV_HALConfig.cmake): configured the way the reference's CMakeLists.txt:925-1043
consumes a HAL -- find_package(OpenCV_HAL NO_MODULE) via -DOpenCV_HAL_DIR, custom_hal.hpp generated from the reference's own template -- and a
program including the reference's imgproc hal_replacement.hpp is built and run against it.  CPU-only: the hooks decline and say so."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "cmake/templates/custom_hal.hpp.in")), reason="needs the reference tree (cmake template + headers)")
@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_hal_package_is_consumable(tmp_path):
    build = tmp_path / "b"
    cfg = os.path.join(ROOT, "oracle", "ref", "cfg")                  # the hand-written cvconfig.h / cv_cpu_config.h the reference headers want
    cmd = ["cmake", "-S", os.path.join(ROOT, "tests", "cmake_hal"), "-B", str(build), f"-DOpenCV_HAL_DIR={os.path.join(ROOT, 'cmake', 'hal')}",
           f"-DOPENCV_SRC={REF}", f"-DCFG_DIR={cfg}", "-DCMAKE_BUILD_TYPE=Release"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mi355cv HAL 0.2.0" in out.stdout
    gen = (build / "gen" / "custom_hal.hpp").read_text()
    assert '#include "mi355cv_hal.hpp"' in gen
    out = subprocess.run(["cmake", "--build", str(build), "-j", "4"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    run = subprocess.run([str(build / "hal_consumer")], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert 'cv_hal_gaussianBlurBinomial -> mi355cv_hal::counted("gaussianBlurBinomial", mi355cv_gaussianBlurBinomial, ARGS)' in run.stdout
    assert 'mi355cv_resize, ARGS)' in run.stdout and 'mi355cv_cvtBGRtoGray, ARGS)' in run.stdout
