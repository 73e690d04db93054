"""The CV_16U sigma-0 Gaussian restated as plain integer binomial sums with one rounding, pinned against the real reference (oracle/_ref, built from
/root/reference by oracle/ref/Makefile) -- TEST INFRASTRUCTURE: this is the checker of tests/test_filters_gpu.py::test_gaussian_16u_binomial_on_the_rolling_kernel."""
import numpy as np
import pytest

import orc


def np_binom16(src, k, border):
    mode = {0: "constant", 1: "edge", 2: "symmetric", 3: "wrap", 4: "reflect"}[border]
    r = k // 2
    p = np.pad(src.astype(np.int64), r, mode=mode)
    t = np.array([1, 2, 1] if k == 3 else [1, 4, 6, 4, 1], np.int64)
    h = sum(t[i] * p[:, i:i + src.shape[1]] for i in range(k))
    v = sum(t[i] * h[i:i + src.shape[0]] for i in range(k))
    return ((v + (8 if k == 3 else 128)) >> (4 if k == 3 else 8)).astype(np.uint16)


def test_16u_binomial_restatement_equals_the_reference():
    if orc.load_ref() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(1)
    for (w, h) in [(37, 23), (64, 5), (5, 64), (333, 41), (1, 9), (9, 1), (2, 2), (3, 3)]:
        src = rng.integers(0, 65536, (h, w)).astype(np.uint16)
        src.flat[:: max(1, src.size // 7)] = 65535
        for k in (3, 5):
            for border in (0, 1, 2, 4):
                want = orc.ref_GaussianBlur(src, k, 0.0, 0.0, border)
                kw = 1 if (w == 1 and border != 0) else k          # smooth.dispatch.cpp:623-630: a one-pixel dimension clamps the kernel
                kh = 1 if (h == 1 and border != 0) else k
                if kw != k or kh != k or min(w, h) < k // 2 + 1:
                    continue                                        # (np.pad cannot reflect beyond the image; the reference's tiny-image branches are not restated here)
                assert np.array_equal(np_binom16(src, k, border), want), (w, h, k, border)
