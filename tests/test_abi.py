"""CPU-only: the C-ABI library loads and exports every symbol include/mi355cv.h declares; the
HAL header maps hooks onto exported symbols; product code never touches the oracle."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "mi355cv.h")).read()
    return sorted(set(re.findall(r"\b(mi355cv_[A-Za-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "opencv_amd", "libmi355cv.so"))
    names = _declared()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_declared_symbols():
    from opencv_amd import _lib
    missing = [n for n in _declared() if n not in _lib.SIGNATURES]
    assert not missing, missing


def test_hal_header_maps_onto_exported_symbols():
    hal = os.path.join(ROOT, "include", "mi355cv_hal.hpp")
    txt = open(hal).read()
    lib = ctypes.CDLL(os.path.join(ROOT, "opencv_amd", "libmi355cv.so"))
    # every binding goes through the counting shim: cv_hal_x(...) -> mi355cv_hal::counted("x", mi355cv_y, ...)
    pairs = re.findall(r"#define\s+(cv_hal_\w+)\(\.\.\.\)\s+mi355cv_hal::counted\(\"\w+\",\s*(mi355cv_\w+),", txt)
    assert len(pairs) >= 65
    for hook, sym in pairs:
        assert hasattr(lib, sym), (hook, sym)
        assert re.search(r"#undef\s+" + hook + r"\b", txt), hook


def test_product_never_references_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "opencv_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                s = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"oracle|libocvref|liboracle", s):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_no_gpu_means_loud_failure_not_fallback():
    import numpy as np
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import opencv_amd as cv
    with pytest.raises(NotImplementedError):
        cv.GaussianBlur(np.zeros((32, 32), np.uint8), 5)
