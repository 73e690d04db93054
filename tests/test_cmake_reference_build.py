"""The reference built by ITS OWN build system with this repository registered as its custom HAL: `oracle/ref/cmake_hal_build.sh` runs
`cmake -S /root/reference -DOpenCV_HAL_DIR=<repo>/cmake/hal ...` + ninja (CMakeLists.txt:946-948, :1033-1039: find_package(OpenCV_HAL NO_MODULE), the
generated custom_hal.hpp includes mi355cv_hal.hpp, libmi355cv.so is linked into the modules) and copies libopencv_*.so + the reference's own
opencv_test_imgproc / opencv_test_video to oracle/_ref/cmake_hal/ (git-ignored, travels to the GPU box).  TEST INFRASTRUCTURE.

 * CPU (here): the generated header names our HAL, libopencv_imgproc.so needs libmi355cv.so, and the reference's bit-exact suites pass on the fallback
   with every hook call tallied as declined.
 * GPU (-m gpu): the suites of the functions behind the hooks, from THAT build's opencv_test_imgproc, pass with the hooks served by the MI355X."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "oracle", "_ref", "cmake_hal")
BIN = os.path.join(OUT, "bin", "opencv_test_imgproc")
pytestmark = pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/cmake_hal not built (bash oracle/ref/cmake_hal_build.sh)")

from test_reference_suite import NEEDS_DATA  # noqa: E402


def run(flt, timeout=1500):
    env = dict(os.environ); env["LD_LIBRARY_PATH"] = os.path.join(OUT, "lib") + ":" + env.get("LD_LIBRARY_PATH", ""); env["MI355CV_PRINT_COUNTS"] = "1"
    p = subprocess.run([BIN, "--gtest_filter=" + flt + "-" + ":".join(NEEDS_DATA), "--gtest_color=no"], cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout, text=True, errors="replace")
    ran = re.search(r"\[==========\] (\d+) tests? from \d+ test (?:cases?|suites?) ran", p.stdout)
    passed = re.search(r"\[  PASSED  \] (\d+) tests?", p.stdout)
    failed = sorted(set(re.findall(r"^\[  FAILED  \] (\S+)", p.stdout, re.M)))
    served = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"^mi355cv: (\S+) (\d+)$", p.stderr, re.M))
    declined = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"^mi355cv: declined (\S+) (\d+)", p.stderr, re.M))
    return p.returncode, int(ran.group(1)) if ran else 0, int(passed.group(1)) if passed else 0, failed, served, declined


def test_the_references_own_cmake_consumed_the_hal_package():
    assert '#include "mi355cv_hal.hpp"' in open(os.path.join(OUT, "custom_hal.hpp")).read()
    assert "Custom HAL" in open(os.path.join(OUT, "configure_summary.txt")).read() and "OpenCV_HAL (ver 0.2.0)" in open(os.path.join(OUT, "configure_summary.txt")).read()
    lib = [f for f in os.listdir(os.path.join(OUT, "lib")) if f.startswith("libopencv_imgproc.so.4.")][0]
    needed = subprocess.run(["readelf", "-d", os.path.join(OUT, "lib", lib)], capture_output=True, text=True).stdout
    assert "libmi355cv.so" in needed


def test_cmake_build_passes_on_the_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: see test_cmake_build_whole_suite_on_the_gpu")
    rc, ran, passed, failed, served, declined = run("GaussianBlur_Bitexact.*:Imgproc_cvtColor_BE.*:Resize_Bitexact.*:Imgproc_Warp*")
    assert rc == 0 and not failed and ran == passed and ran >= 50, (rc, ran, passed, failed[:10])
    assert not served and declined.get("gaussianBlurBinomial", 0) > 0 and declined.get("warpAffine", 0) > 0, (served, declined)


# the suites of the functions behind the hooks (the scope of oracle/ref's own reduced build of the test binary, tests/test_reference_suite.py): the cmake build's
# binary also carries the reference's contour / histogram-comparison / Hough / ... suites, minutes of CPU-only work that no hook touches
HOOK_SUITES = ("GaussianBlur_Bitexact.*:Resize_Bitexact.*:Imgproc_Erode.*:Imgproc_Dilate.*:Imgproc_MorphologyEx.*:Imgproc_Filter2D.*:Imgproc_Sobel.*:Imgproc_SpatialGradient.*:"
               "Imgproc_Laplace.*:Imgproc_Blur.*:Imgproc_GaussianBlur.*:Imgproc_MedianBlur.*:Imgproc_PyramidDown.*:Imgproc_PyramidUp.*:Imgproc_MinEigenVal.*:Imgproc_EigenValsVecs.*:"
               "Imgproc_PreCornerDetect.*:Imgproc_Integral.*:Imgproc_Morphology.*:Imgproc_MorphEx.*:Imgproc_Pyrdown.*:Imgproc_Color*:Imgproc_cvtColor_BE.*:ImgProc_RGB2YUV.*:"
               "ImgProc_cvtColorTwoPlane.*:ImgProc_RGB2Lab.*:cvtColorUYVY.*:Imgproc_cvWarpAffine.*:Imgproc_resize_area.*:Imgproc_Resize*:Imgproc_WarpAffine*:Imgproc_WarpPerspective*:"
               "Imgproc_Remap*:Resize.*:Imgproc_Warp.*:Imgproc_linearPolar.*:Imgproc_logPolar.*:Imgproc_warpPolar.*:Imgproc_Threshold.*:Imgproc_PyrUp.*:Imgproc_MatchTemplate.*:"
               "Imgproc_BilateralFilter.*:Imgproc_Moments.*:cvt420/*:cvt422/*:matchTemplate_Modes.*:Imgproc_Hist/Imgproc_Equalize_Hist.*:Imgproc_FilterSupportedFormats.*")


@pytest.mark.gpu
def test_cmake_build_hook_suites_on_the_gpu():
    rc, ran, passed, failed, served, declined = run(HOOK_SUITES, timeout=600)
    assert rc == 0 and not failed and ran == passed and ran > 250, (rc, ran, passed, failed[:10])
    for hook in ("gaussianBlurBinomial", "filter", "sepFilter", "boxFilter", "resize", "warpAffine", "warpPerspective", "cvtBGRtoGray", "pyrdown", "threshold", "integral"):
        assert served.get(hook, 0) > 0, (hook, served)
