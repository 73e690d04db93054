"""Pins oracle/lk.c (cv_hal_ScharrDeriv / cv_hal_LKOpticalFlowLevel granularity) against the real reference's
cv::calcOpticalFlowPyrLK (CPU only): next points, status and error BIT-EXACT, for window widths that exercise the reference's vector
body, its scalar tail and both, 1 and 3 channels, all flag combinations, points at and beyond the image edge."""
import numpy as np
import pytest

import orc as o

pytestmark = pytest.mark.skipif(o.load_ref() is None, reason="oracle/_ref/libocvref.so not built")


def frames(h, w, cn, seed, shift=(2.3, -1.6)):
    """a smooth random texture and a sub-pixel shifted, slightly brighter copy"""
    rng = np.random.default_rng(seed)
    base = rng.random((h // 8 + 3, w // 8 + 3, cn))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)

    def sample(dx, dy):
        x = (xx + dx) / 8.0 + 1; y = (yy + dy) / 8.0 + 1
        x0 = np.floor(x).astype(int); y0 = np.floor(y).astype(int)
        fx = (x - x0)[..., None]; fy = (y - y0)[..., None]
        x0 = np.clip(x0, 0, base.shape[1] - 2); y0 = np.clip(y0, 0, base.shape[0] - 2)
        v = base[y0, x0] * (1 - fx) * (1 - fy) + base[y0, x0 + 1] * fx * (1 - fy) + base[y0 + 1, x0] * (1 - fx) * fy + base[y0 + 1, x0 + 1] * fx * fy
        return v
    a = sample(0, 0); b = sample(*shift) * 0.97 + 0.01
    noise = rng.normal(0, 0.01, a.shape)
    A = np.clip(a * 255, 0, 255).astype(np.uint8); B = np.clip((b + noise) * 255, 0, 255).astype(np.uint8)
    if cn == 1:
        A, B = A[..., 0], B[..., 0]
    return np.ascontiguousarray(A), np.ascontiguousarray(B)


def points(h, w, n, seed):
    rng = np.random.default_rng(seed)
    p = np.stack([rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)], axis=1).astype(np.float32)
    p[:4] = [[0, 0], [w - 1, h - 1], [w + 30.5, 10], [10.25, -40]]
    return p


def same(a, b, what):
    for x, y, name in zip(a, b, ("nextPts", "status", "err")):
        if name == "nextPts":                      # a lost point keeps whatever the last level left there in both implementations
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (what, name, int((x != y).any(axis=1).sum()))
        else:
            assert np.array_equal(x, y), (what, name)


def test_scharr_deriv_via_full_pipeline_single_level():
    """maxLevel 0: no pyrDown involved -- isolates ScharrDeriv + one tracker level"""
    for cn in (1, 3):
        A, B = frames(120, 160, cn, 1)
        p = points(120, 160, 200, 2)
        same(o.orc_calcOpticalFlowPyrLK(A, B, p, (21, 21), 0), o.ref_calcOpticalFlowPyrLK(A, B, p, (21, 21), 0), ("level0", cn))


@pytest.mark.parametrize("cn", [1, 3])
@pytest.mark.parametrize("win", [(21, 21), (15, 15), (9, 11), (8, 8), (5, 7), (31, 13)])
def test_pyr_lk_windows(cn, win):
    A, B = frames(240, 320, cn, 3 + cn)
    p = points(240, 320, 300, 5)
    same(o.orc_calcOpticalFlowPyrLK(A, B, p, win, 3), o.ref_calcOpticalFlowPyrLK(A, B, p, win, 3), (win, cn))


def test_pyr_lk_flags_and_criteria():
    A, B = frames(200, 260, 1, 9, shift=(5.2, 3.1))
    p = points(200, 260, 250, 6)
    guess = p + np.float32([4.0, 2.5])
    for flags, nextPts in ((0, None), (8, None), (4, guess), (12, guess)):
        for crit in ((3, 30, 0.01), (1, 5, 0.0), (2, 0, 0.3), (3, 0, 0.01), (0, 0, 0.0), (3, 100, 1e-4)):
            for minEig in (1e-4, 1e-2):
                got = o.orc_calcOpticalFlowPyrLK(A, B, p, (21, 21), 2, crit, flags, minEig, nextPts)
                want = o.ref_calcOpticalFlowPyrLK(A, B, p, (21, 21), 2, crit, flags, minEig, nextPts)
                same(got, want, (flags, crit, minEig))


def test_pyr_lk_level_cap_and_flat_image():
    A, B = frames(70, 90, 1, 11)
    p = points(70, 90, 60, 7)
    same(o.orc_calcOpticalFlowPyrLK(A, B, p, (21, 21), 5), o.ref_calcOpticalFlowPyrLK(A, B, p, (21, 21), 5), "cap")     # pyramid stops early (:836)
    flat = np.full((100, 100), 77, np.uint8)
    same(o.orc_calcOpticalFlowPyrLK(flat, flat, p, (11, 11), 2), o.ref_calcOpticalFlowPyrLK(flat, flat, p, (11, 11), 2), "flat")
    st = o.orc_calcOpticalFlowPyrLK(A, B, points(70, 90, 60, 8), (21, 21), 2)[1]
    assert 0 < st.sum() < len(st)                                                # the test set does contain both tracked and lost points
