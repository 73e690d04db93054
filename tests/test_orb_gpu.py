"""GPU parity for SURVEY §8 f3 "features2d detectors": cv::ORB (detect / compute / detectAndCompute) through the C ABI against the restatement
pinned to the reference's orb.cpp (tests/test_oracle_orb.py): keypoints byte for byte -- coordinates, size, angle, response, octave AND their
order -- and descriptors bit for bit.  The reference's own checks for ORB (modules/features2d/test/test_orb.cpp) ask for less: a keypoint count,
image-border distances, and repeatable descriptors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

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
    (1001, 37, 10, dict(edgeThreshold=4, patchSize=7, nlevels=4)),      # a strip: levels wrap in the buffer
]


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def dev(a):
    return torch.from_numpy(a).cuda()


def same(got, want):
    gk, gd = got
    wk, wd = want
    assert len(gk) == len(wk), (len(gk), len(wk))
    for f in wk.dtype.names:
        assert np.array_equal(gk[f].view(np.int32), wk[f].view(np.int32)), f
    if wd is None:
        assert gd is None
    else:
        assert gd.shape == wd.shape and np.array_equal(gd, wd)


@pytest.mark.parametrize("w,h,seed,kw", CASES)
def test_orb_detect_and_compute(cv, orc, w, h, seed, kw):
    img = orc.orb_scene(w, h, seed)
    want = orc.orc_ORB(img, **kw)
    n0 = cv.call_count("ORB_detectAndCompute")
    orb = cv.ORB_create(**kw)
    same(orb.detectAndCompute(dev(img)), want)
    same(orb.detectAndCompute(img), want)                                # host image: staged
    assert cv.call_count("ORB_detectAndCompute") == n0 + 2
    k = orb.detect(dev(img))
    assert k.tobytes() == want[0].tobytes()


def test_orb_mask(cv, orc):
    for (w, h, seed, kw) in [(640, 480, 20, {}), (400, 300, 21, dict(firstLevel=1, nfeatures=800)), (500, 375, 22, dict(scoreType=1, scaleFactor=1.4, nlevels=6))]:
        img, mask = orc.orb_scene(w, h, seed), orc.orb_mask(w, h, seed)
        want = orc.orc_ORB(img, mask=mask, **kw)
        orb = cv.ORB_create(**kw)
        same(orb.detectAndCompute(dev(img), dev(mask)), want)
        same(orb.detectAndCompute(img, mask), want)
        assert want[0].tobytes() != orc.orc_ORB(img, **kw)[0].tobytes()
    with pytest.raises(ValueError):
        orb.detectAndCompute(dev(img), dev(mask[:-1]))


def test_orb_compute_on_provided_keypoints(cv, orc):
    img = orc.orb_scene(480, 360, 11)
    orb = cv.ORB_create()
    k, _ = orc.orc_ORB(img, descriptors=False)
    rng = np.random.default_rng(5)
    kp = k[rng.permutation(len(k))[:200]].copy()                          # out of level order, some too close to the edge, some off the pixel grid
    kp["x"][:5] = 3.0
    kp["x"][5:10] += 0.37
    want = orc.orc_ORB(img, keypoints=kp)
    same(orb.compute(dev(img), kp), want)
    same(orb.compute(img, kp), want)
    assert 150 < len(want[0]) < 200
    # nothing to describe
    gk, gd = orb.compute(dev(img), kp[:0])
    assert len(gk) == 0 and gd.shape == (0, 32)


def test_orb_colour_input_and_views(cv, orc):
    rng = np.random.default_rng(2)
    g = orc.orb_scene(400, 300, 31)
    bgr = np.stack([g, np.roll(g, 3, 1), np.roll(g, 5, 0)], axis=-1).copy()
    gray = cv.cvtColor(bgr, cv.COLOR_BGR2GRAY)
    orb = cv.ORB_create(nfeatures=400)
    same(orb.detectAndCompute(dev(bgr)), orc.orc_ORB(np.ascontiguousarray(gray), nfeatures=400))
    # a window of a larger device frame (row stride > width)
    big = rng.integers(0, 256, (340, 512), dtype=np.uint8)
    big[20:320, 40:440] = g
    same(orb.detectAndCompute(dev(big)[20:320, 40:440]), orc.orc_ORB(g, nfeatures=400))


def test_orb_full_hd_frame(cv, orc):
    """1080p, 5000 features, 8 levels: every stage at a size where levels wrap into rows of the buffer and candidate lists are long"""
    img = orc.orb_scene(1920, 1080, 40)
    kw = dict(nfeatures=5000)
    want = orc.orc_ORB(img, **kw)
    assert len(want[0]) > 3000
    same(cv.ORB_create(**kw).detectAndCompute(dev(img)), want)


def test_orb_setters_and_the_reference_regressions(cv, orc):
    """setScaleFactor keeps a double where ORB::create rounds to float (orb.cpp:660, :1262); the geometries of the reference's own ORB tests that need no
    image file (modules/features2d/test/test_orb.cpp: crash_5031 -- compute() on a keypoint 5 rows from the top with an 18-level, patch-47 detector;
    regression_16197 -- firstLevel 3, scale 1.8, patch 8)"""
    img = orc.orb_scene(300, 220, 31)
    orb = cv.ORB_create(nfeatures=900, nlevels=5)
    orb.setScaleFactor(1.8)
    assert orb.getScaleFactor() == 1.8
    same(orb.detectAndCompute(dev(img)), orc.orc_ORB(img, setScaleFactor=1.8, nlevels=5, nfeatures=900))
    same(cv.ORB_create(nfeatures=900, nlevels=5, scaleFactor=1.8).detectAndCompute(dev(img)), orc.orc_ORB(img, scaleFactor=1.8, nlevels=5, nfeatures=900))
    small = orc.orb_scene(72, 72, 16197)
    o2 = cv.ORB_create()
    o2.setNLevels(5); o2.setFirstLevel(3); o2.setScaleFactor(1.8); o2.setPatchSize(8); o2.setEdgeThreshold(8)
    same(o2.detectAndCompute(dev(small)), orc.orc_ORB(small, nlevels=5, firstLevel=3, setScaleFactor=1.8, patchSize=8, edgeThreshold=8))
    k, d = o2.detectAndCompute(dev(np.zeros((72, 72), np.uint8)))                                # the reference's own input: all zeros
    assert len(k) == 0 and d.shape == (0, 32)
    o3 = cv.ORB_create(8000, 1.2, 18, 4, 0, 2, cv.ORB_HARRIS_SCORE, 47, 20)
    kp = np.zeros(1, cv.KEYPOINT_DTYPE)
    kp["x"], kp["y"], kp["size"], kp["angle"], kp["response"], kp["octave"], kp["class_id"] = 443, 5, 47, 53.4580612, 0.0000470733867, 0, -1
    bgr = np.zeros((1080, 1920, 3), np.uint8)
    gk, gd = o3.compute(dev(bgr), kp)
    wk, wd = orc.orc_ORB(np.zeros((1080, 1920), np.uint8), keypoints=kp, nfeatures=8000, nlevels=18, edgeThreshold=4, patchSize=47)
    assert gk.tobytes() == wk.tobytes() and np.array_equal(gd, wd) and len(gk) == 1


def test_orb_declines_and_errors(cv, orc):
    img = dev(orc.orb_scene(320, 240, 1))
    with pytest.raises(ValueError):
        cv.ORB_create(patchSize=1).detect(img)
    with pytest.raises(ValueError):
        cv.ORB_create(firstLevel=-1)
    with pytest.raises(NotImplementedError):
        cv.ORB_create(nlevels=cv.limit("orb_max_levels") + 1).detect(img)
    with pytest.raises(ValueError):
        cv.ORB_create().detect(img.to(torch.float32))
    k, d = cv.ORB_create().detectAndCompute(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d is None
    # a flat image: no keypoints, no descriptors
    k, d = cv.ORB_create().detectAndCompute(dev(np.full((240, 320), 77, np.uint8)))
    assert len(k) == 0 and d.shape == (0, 32)


def test_orb_frames_across_host_threads(cv, orc):
    """frames of a video on several host threads (per-thread streams, scratch pools and page-locked landing zones inside the library; ctypes releases
    the GIL for the call): one call's host round trips and culls overlap the other calls' kernels.  Every frame's result equals the restatement's."""
    import threading
    frames = [orc.orb_scene(640, 480, 50 + i) for i in range(6)]
    want = [orc.orc_ORB(f, nfeatures=800) for f in frames]
    dev_frames = [dev(f) for f in frames]
    torch.cuda.synchronize()
    errors = []

    def worker(tid):
        try:
            orb = cv.ORB_create(nfeatures=800)
            for rep in range(4):
                for i in range(tid, len(frames), 3):
                    k, d = orb.detectAndCompute(dev_frames[i] if (rep + tid) % 2 else frames[i])
                    if k.tobytes() != want[i][0].tobytes() or not np.array_equal(d, want[i][1]):
                        errors.append((tid, rep, i))
        except Exception as e:                      # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_orb_random_parameter_sets(cv, orc):
    """the thirty seeded random parameter sets of tests/test_oracle_orb.py (restatement == reference on each) through the GPU path"""
    rng = np.random.default_rng(2024)
    for t in range(30):
        w, h = int(rng.integers(120, 420)), int(rng.integers(90, 320))
        nl = int(rng.integers(1, 9)); fl = int(rng.integers(0, min(3, nl)))
        kw = dict(nfeatures=int(rng.integers(50, 1500)), scaleFactor=float(np.round(rng.uniform(1.1, 2.0), 2)), nlevels=nl, edgeThreshold=int(rng.integers(3, 32)), firstLevel=fl,
                  WTA_K=int(rng.choice([2, 3, 4])), scoreType=int(rng.integers(0, 2)), patchSize=int(rng.integers(5, 32)), fastThreshold=int(rng.integers(5, 40)))
        img = orc.orb_scene(w, h, 100 + t)
        same(cv.ORB_create(**kw).detectAndCompute(dev(img)), orc.orc_ORB(img, **kw))
