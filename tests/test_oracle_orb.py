"""The ORB restatement (oracle/orb.c) pinned against the reference itself (oracle/_ref: modules/features2d/src/orb.cpp compiled from
/root/reference): keypoints byte for byte -- coordinates, size, angle, response, octave AND order (the order is libstdc++'s nth_element) --
and descriptors.  SURVEY section 8 f3."""
import ctypes

import numpy as np
import pytest

import orc as o

needs_ref = pytest.mark.skipif(o.load_ref() is None, reason="oracle/_ref not built (needs /root/reference)")

CASES = [
    (640, 480, 0, {}),
    (320, 240, 1, dict(nfeatures=300)),
    (800, 600, 2, dict(nfeatures=1500, scaleFactor=1.5, nlevels=5)),
    (500, 375, 3, dict(scoreType=1)),                                   # FAST_SCORE: ties at the cut are all kept
    (640, 480, 4, dict(WTA_K=3)),
    (640, 480, 5, dict(WTA_K=4, edgeThreshold=19, patchSize=19)),       # random pattern (patchSize != 31)
    (400, 300, 6, dict(firstLevel=1)),                                  # level 0 is an upscale
    (333, 222, 7, dict(edgeThreshold=5, nfeatures=2000, fastThreshold=5)),   # descriptors reach into the reflected border
    (97, 61, 8, dict(nlevels=3, edgeThreshold=8, patchSize=9)),
    (64, 48, 9, {}),                                                    # nothing survives the 31-pixel edge on the small levels
    (640, 480, 12, dict(scaleFactor=2.0, nlevels=4, nfeatures=800)),    # exactly-half levels: INTER_LINEAR_EXACT becomes the 2 x 2 mean (resize.cpp:3976)
    (641, 479, 13, dict(scaleFactor=2.0, nlevels=3, firstLevel=1)),     # and an exact 2 x upscale
]


@needs_ref
@pytest.mark.parametrize("w,h,seed,kw", CASES)
def test_orb_restatement_equals_reference(w, h, seed, kw):
    img = o.orb_scene(w, h, seed)
    rk, rd = o.ref_ORB(img, **kw)
    ok, od = o.orc_ORB(img, **kw)
    assert len(rk) == len(ok)
    for f in o.KP_DTYPE.names:
        assert np.array_equal(rk[f].view(np.int32), ok[f].view(np.int32)), f
    assert np.array_equal(rd, od)


@needs_ref
@pytest.mark.parametrize("w,h,seed,kw", [(640, 480, 20, {}), (400, 300, 21, dict(firstLevel=1, nfeatures=800)), (500, 375, 22, dict(scoreType=1, scaleFactor=1.4, nlevels=6))])
def test_orb_with_a_mask(w, h, seed, kw):
    """the mask pyramid (resize + THRESH_TOZERO at 254 per level) and KeyPointsFilter::runByPixelsMask after FAST"""
    img, mask = o.orb_scene(w, h, seed), o.orb_mask(w, h, seed)
    rk, rd = o.ref_ORB(img, mask=mask, **kw)
    ok, od = o.orc_ORB(img, mask=mask, **kw)
    fk, _ = o.ref_ORB(img, **kw)
    assert rk.tobytes() == ok.tobytes() and np.array_equal(rd, od)
    assert len(rk) > 50 and rk.tobytes() != fk.tobytes()
    assert np.all(mask[np.rint(rk["y"][rk["octave"] == kw.get("firstLevel", 0)]).astype(int), np.rint(rk["x"][rk["octave"] == kw.get("firstLevel", 0)]).astype(int)] != 0)


@needs_ref
def test_orb_random_parameter_sets():
    """thirty seeded draws of every ORB::create parameter (image size, feature budget, scale factor, levels, first level, edge threshold, patch size, WTA_K,
    score type, FAST threshold): the restatement equals the reference on each -- keypoints, their order, descriptors"""
    rng = np.random.default_rng(2024)
    for t in range(30):
        w, h = int(rng.integers(120, 420)), int(rng.integers(90, 320))
        nl = int(rng.integers(1, 9)); fl = int(rng.integers(0, min(3, nl)))
        kw = dict(nfeatures=int(rng.integers(50, 1500)), scaleFactor=float(np.round(rng.uniform(1.1, 2.0), 2)), nlevels=nl, edgeThreshold=int(rng.integers(3, 32)), firstLevel=fl,
                  WTA_K=int(rng.choice([2, 3, 4])), scoreType=int(rng.integers(0, 2)), patchSize=int(rng.integers(5, 32)), fastThreshold=int(rng.integers(5, 40)))
        img = o.orb_scene(w, h, 100 + t)
        a, b = o.ref_ORB(img, **kw), o.orc_ORB(img, **kw)
        assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]), (t, w, h, kw)


@needs_ref
def test_orb_scale_factor_set_as_a_double():
    """ORB::create takes the scale factor as a float, setScaleFactor as a double (orb.cpp:660, :1262): 1.8 and float(1.8) give different level sizes.  Also the
    geometry of the reference's own regression_16197 (test_orb.cpp:127: firstLevel 3, 1.8, patch 8, edge 8 -- level 0 is a 5.8 x upscale) on a scene with corners"""
    img = o.orb_scene(300, 220, 31)
    a = o.ref_ORB(img, setScaleFactor=1.8, nlevels=5, nfeatures=900)
    b = o.orc_ORB(img, setScaleFactor=1.8, nlevels=5, nfeatures=900)
    c = o.ref_ORB(img, scaleFactor=1.8, nlevels=5, nfeatures=900)
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and len(a[0]) > 200
    assert a[0].tobytes() != c[0].tobytes()                                  # the float and the double differ in the output
    small = o.orb_scene(72, 72, 16197)
    kw = dict(nlevels=5, firstLevel=3, setScaleFactor=1.8, patchSize=8, edgeThreshold=8)
    ra, oa = o.ref_ORB(small, **kw), o.orc_ORB(small, **kw)
    assert ra[0].tobytes() == oa[0].tobytes() and np.array_equal(ra[1], oa[1]) and len(ra[0]) > 20


@needs_ref
def test_orb_detect_only_and_provided_keypoints():
    img = o.orb_scene(480, 360, 11)
    rk, _ = o.ref_ORB(img, descriptors=False)
    ok, _ = o.orc_ORB(img, descriptors=False)
    assert rk.tobytes() == ok.tobytes() and len(rk) > 100
    # compute(): keypoints handed in, out of level order, some too close to the edge
    rng = np.random.default_rng(5)
    kp = rk[rng.permutation(len(rk))[:200]].copy()
    kp["x"][:5] = 3.0
    kp["x"][5:10] += 0.37
    rk2, rd2 = o.ref_ORB(img, keypoints=kp)
    ok2, od2 = o.orc_ORB(img, keypoints=kp)
    assert rk2.tobytes() == ok2.tobytes() and np.array_equal(rd2, od2) and 150 < len(rk2) < 200


@needs_ref
def test_retain_best_is_libstdcxx_nth_element():
    rng = np.random.default_rng(3)
    r, c = o.load_ref(), o.oracle()
    for trial in range(300):
        n = int(rng.integers(1, 600))
        kp = np.zeros(n, o.KP_DTYPE)
        kp["x"] = np.arange(n)
        mode = trial % 4
        if mode == 0: kp["response"] = rng.integers(0, 12, n)                  # heavy ties (FAST scores)
        elif mode == 1: kp["response"] = rng.random(n)
        elif mode == 2: kp["response"] = np.sort(rng.integers(0, 200, n))[::-1]
        else: kp["response"] = np.where(rng.random(n) < 0.9, 7, rng.integers(0, 20, n))
        for npts in {0, 1, n // 3, n // 2, max(n - 1, 0), n, n + 5}:
            a, b = kp.copy(), kp.copy()
            na = r.ref_retainBest(o.P(a), n, npts)
            nb = c.orc_retainBest(o.P(b), n, npts)
            assert na == nb and a[:na].tobytes() == b[:nb].tobytes(), (trial, n, npts)


@needs_ref
def test_retain_best_heap_select_branch():
    """introselect falls back to heap selection when median-of-three keeps choosing badly: an organ-pipe killer sequence"""
    r, c = o.load_ref(), o.oracle()
    for n in (64, 257, 1000, 4096):
        # median-of-3 killer (Musser): forces the depth limit
        k = n // 2
        v = np.zeros(n)
        for i in range(1, k + 1):
            if i % 2 == 1: v[i - 1] = i; v[i] = k + i
            v[k + i - 1] = 2 * i
        kp = np.zeros(n, o.KP_DTYPE)
        kp["response"] = -v[:n]
        kp["x"] = np.arange(n)
        for npts in (n // 2, n // 3, n - 2):
            a, b = kp.copy(), kp.copy()
            na = r.ref_retainBest(o.P(a), n, npts)
            nb = c.orc_retainBest(o.P(b), n, npts)
            assert na == nb and a[:na].tobytes() == b[:nb].tobytes(), (n, npts)
    assert c.orc_heapSelectCalls() > 0          # the sequence really exhausted the depth budget


@needs_ref
def test_fast_atan2():
    r, c = o.load_ref(), o.oracle()
    r.ref_fastAtan2.restype = ctypes.c_float; r.ref_fastAtan2.argtypes = [ctypes.c_float, ctypes.c_float]
    c.orc_fastAtan2.restype = ctypes.c_float; c.orc_fastAtan2.argtypes = [ctypes.c_float, ctypes.c_float]
    rng = np.random.default_rng(0)
    ys = np.concatenate([rng.integers(-200000, 200000, 20000), [0, 0, 1, -1, 5, 0]]).astype(np.float32)
    xs = np.concatenate([rng.integers(-200000, 200000, 20000), [0, 3, 0, 0, 5, -7]]).astype(np.float32)
    for y, x in zip(ys, xs):
        a, b = r.ref_fastAtan2(y, x), c.orc_fastAtan2(y, x)
        assert np.float32(a).tobytes() == np.float32(b).tobytes(), (y, x, a, b)
