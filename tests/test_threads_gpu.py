"""Re-entrancy (SURVEY §8b "Threading"): cv:: functions may be called concurrently from many host threads, so the hooks keep
per-thread streams / staging pools.  Eight threads hammer different hooks on host arrays (staged) and device tensors at once;
every result must equal the oracle's."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_concurrent_hooks(orc):
    import opencv_amd as cv
    rng = np.random.default_rng(99)
    img = rng.integers(0, 256, (240, 320, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (240, 320), dtype=np.uint8)
    k = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float32)
    M = cv.getRotationMatrix2D((160.0, 120.0), 11.0, 0.9)
    want = {
        "gauss": orc.orc_gaussianBlurBinomialU8(img, 5, 4),
        "filter": orc.orc_filter2D(gray, -1, k),
        "box": orc.orc_boxFilter(img, -1, (5, 5)),
        "sobel": orc.orc_Sobel(gray, 3, 1, 0, 3),
        "gray": orc.orc_cvtColor(img, 6),
        "resize": orc.orc_resize(img, (200, 150)),
        "warp": orc.orc_warpAffine(img, M, (320, 240)),
        "dilate": orc.orc_morph(1, gray),
        "thresh": orc.orc_threshold(gray, 100, 255, 0)[1],
        "harris": orc.orc_cornerHarris(gray, 2, 3, 0.04),
    }
    calls = {
        "gauss": lambda a, g: cv.GaussianBlur(a, (5, 5), 0),
        "filter": lambda a, g: cv.filter2D(g, -1, k),
        "box": lambda a, g: cv.boxFilter(a, -1, (5, 5)),
        "sobel": lambda a, g: cv.Sobel(g, cv.CV_16S, 1, 0, 3),
        "gray": lambda a, g: cv.cvtColor(a, cv.COLOR_BGR2GRAY),
        "resize": lambda a, g: cv.resize(a, (200, 150)),
        "warp": lambda a, g: cv.warpAffine(a, M, (320, 240), cv.INTER_LINEAR | cv.WARP_INVERSE_MAP),
        "dilate": lambda a, g: cv.dilate(g),
        "thresh": lambda a, g: cv.threshold(g, 100, 255, 0)[1],
        "harris": lambda a, g: cv.cornerHarris(g, 2, 3, 0.04),
    }
    errors = []

    def worker(tid):
        try:
            names = list(calls)
            dimg, dgray = (torch.from_numpy(img).cuda(), torch.from_numpy(gray).cuda()) if tid % 2 else (img, gray)
            for it in range(12):
                n = names[(tid + it) % len(names)]
                got = calls[n](dimg, dgray)
                got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
                if got.dtype == np.float32:
                    import orc as O
                    ok = O.rel_err(got, want[n]) <= 1e-4
                else:
                    ok = np.array_equal(got, want[n])
                if not ok:
                    errors.append((tid, it, n))
        except Exception as e:                      # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
