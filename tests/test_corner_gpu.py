"""GPU parity for rows a10-a12: cornerHarris / cornerMinEigenVal (fused LDS kernel), goodFeaturesToTrack, pyrDown /
buildPyramid -- through the C ABI against the oracle.  Float responses: 1e-4 relative in the reference's global norm
(test/ocl/test_imgproc.cpp:246-263); pyrDown integers and gftt corner lists: exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def dev(a):
    return torch.from_numpy(a).cuda()


def structured(h, w, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    import orc
    img = orc.orc_gaussianBlurBinomialU8(orc.orc_gaussianBlurBinomialU8(img, 9, 4), 9, 4).copy()
    img[h // 4:h // 2, w // 3:w // 3 + w // 5] = 220
    img[h // 2 + 5:h // 2 + 25, w // 8:w // 8 + 30] = 30
    return img


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_corner_response(cv, orc, dtype):
    for (w, h) in [(64, 48), (133, 77), (65, 17), (7, 5)]:
        img = structured(max(h, 60), max(w, 80), 5 + w)[:h, :w].copy()
        src = img if dtype == np.uint8 else (img.astype(np.float32) / 255.0)
        for bs, ks in [(2, 3), (3, 3), (5, 5), (2, 7), (4, -1), (1, 3), (2, 1)]:
            for border in (0, 1, 2, 4):
                want = orc.orc_cornerHarris(src, bs, ks, 0.04, border)
                got = cv.cornerHarris(dev(src), bs, ks, 0.04, border).cpu().numpy()
                assert orc.rel_err(got, want) <= 1e-4, (w, h, bs, ks, border)
                if bs > 1:
                    want = orc.orc_cornerMinEigenVal(src, bs, ks, border)
                    got = cv.cornerMinEigenVal(dev(src), bs, ks, border).cpu().numpy()
                    assert orc.rel_err(got, want) <= 1e-4, (w, h, bs, ks, border)
    img = structured(120, 160, 3)
    assert orc.rel_err(cv.cornerHarris(img, 2, 3, 0.04), orc.orc_cornerHarris(img, 2, 3, 0.04)) <= 1e-4     # host arrays


def test_corner_rolling_path(cv, orc):
    """blockSize 2 / ksize 3 / CV_8UC1 with W % 8 == 0 takes the register-rolling kernel: both responses, every border mode,
    heights that give one / several / ragged segments (walking up and down), several strips; within 1e-4 of the oracle;
    against the LDS-tiled kernel (double box sums like the reference) MinEigenVal is bit-identical and Harris (float box
    sums) within 1e-5."""
    import os
    rng = np.random.default_rng(17)
    for (w, h) in [(16, 2), (8, 5), (24, 3), (64, 48), (520, 37), (1048, 37), (2064, 101), (1920, 270)]:
        img = structured(max(h, 60), max(w, 80), 5 + w)[:h, :w].copy() if w < 2000 else rng.integers(0, 256, (h, w), dtype=np.uint8)
        for border in (0, 1, 2, 4):
            want = orc.orc_cornerHarris(img, 2, 3, 0.04, border)
            got = cv.cornerHarris(dev(img), 2, 3, 0.04, border).cpu().numpy()
            assert orc.rel_err(got, want) <= 1e-4, (w, h, border)
            wantm = orc.orc_cornerMinEigenVal(img, 2, 3, border)
            gotm = cv.cornerMinEigenVal(dev(img), 2, 3, border).cpu().numpy()
            assert orc.rel_err(gotm, wantm) <= 1e-4, (w, h, border)
            os.environ["MI355CV_CORNER_LDS"] = "1"
            try:
                lds = cv.cornerHarris(dev(img), 2, 3, 0.04, border).cpu().numpy()
                ldsm = cv.cornerMinEigenVal(dev(img), 2, 3, border).cpu().numpy()
            finally:
                del os.environ["MI355CV_CORNER_LDS"]
            assert orc.rel_err(got, lds) <= 1e-5 and np.array_equal(gotm, ldsm), (w, h, border, orc.rel_err(got, lds), orc.rel_err(gotm, ldsm))
    frames = rng.integers(0, 256, (5, 64, 96), dtype=np.uint8)
    resp = cv.cornerHarrisBatch(dev(frames), 2, 3, 0.04).cpu().numpy()
    for i in range(5):
        assert orc.rel_err(resp[i], orc.orc_cornerHarris(frames[i], 2, 3, 0.04)) <= 1e-4


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32])
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_pyrdown(cv, orc, dtype, cn):
    rng = np.random.default_rng(cn)
    for (w, h) in [(64, 48), (65, 49), (31, 7), (2, 2), (5, 1), (1, 9)]:
        shape = (h, w, cn) if cn > 1 else (h, w)
        src = rng.random(shape, dtype=np.float32) if dtype == np.float32 else rng.integers(np.iinfo(dtype).min, int(np.iinfo(dtype).max) + 1, shape, dtype=dtype)
        for border in (1, 2, 3, 4):
            want = orc.orc_pyrDown(src, None, border)
            got = cv.pyrDown(dev(src), None, border).cpu().numpy()
            assert (orc.rel_err(got, want) <= 1e-6) if dtype == np.float32 else np.array_equal(got, want), (w, h, border)
    src = rng.integers(0, 256, (48, 64, cn) if cn > 1 else (48, 64)).astype(dtype)
    for dsize in [(33, 25), (31, 23)]:
        want = orc.orc_pyrDown(src, dsize, 4)
        got = cv.pyrDown(dev(src), dsize).cpu().numpy()
        assert (orc.rel_err(got, want) <= 1e-6) if dtype == np.float32 else np.array_equal(got, want)
    parent = rng.integers(0, 256, (60, 90, cn) if cn > 1 else (60, 90)).astype(dtype)
    x0, y0, w, h = 10, 8, 40, 30
    want = orc.orc_pyrDown(parent[y0:y0 + h, x0:x0 + w], None, 4, margins=(x0, y0, 90 - x0 - w, 60 - y0 - h))
    got = cv.pyrDown(dev(parent)[y0:y0 + h, x0:x0 + w], None, 4, margins=(x0, y0, 90 - x0 - w, 60 - y0 - h)).cpu().numpy()
    assert (orc.rel_err(got, want) <= 1e-6) if dtype == np.float32 else np.array_equal(got, want)


def test_pyrdown_rolling_path(cv, orc):
    """CV_8UC1 with W % 16 == 0 and the default destination size takes the register-rolling kernel."""
    rng = np.random.default_rng(23)
    for (w, h) in [(16, 2), (16, 3), (32, 7), (64, 48), (1040, 37), (2064, 101), (1920, 135), (240, 135)]:
        src = rng.integers(0, 256, (h, w), dtype=np.uint8)
        for border in (1, 2, 4):
            assert np.array_equal(cv.pyrDown(dev(src), None, border).cpu().numpy(), orc.orc_pyrDown(src, None, border)), (w, h, border)
    fr = rng.integers(0, 256, (5, 70, 96), dtype=np.uint8)
    got = cv.buildPyramidBatch(dev(fr), 2)
    for i in range(5):
        l1 = orc.orc_pyrDown(fr[i]); l2 = orc.orc_pyrDown(l1)
        assert np.array_equal(got[1][i].cpu().numpy(), l1) and np.array_equal(got[2][i].cpu().numpy(), l2)


def test_build_pyramid_fused_levels(cv, orc, monkeypatch):
    """buildPyramidBatch with maxlevel >= 4 produces its last three levels in one launch (k_pyr3): odd and even sizes at every level, sizes
    smaller than one tile, several tiles with ragged edges, three border rules, views with a parent's step; equal to the level-by-level
    launches (MI355CV_PYR_FUSE=0) and to the oracle"""
    rng = np.random.default_rng(77)
    for (w, h, n, ml) in [(1920, 1080, 2, 4), (640, 480, 3, 4), (512, 512, 1, 5), (333, 257, 2, 4), (130, 70, 2, 4), (1040, 1030, 1, 5), (96, 800, 2, 4)]:
        fr = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        d = dev(fr)
        for border in (4, 1, 2):
            got = cv.buildPyramidBatch(d, ml, border)
            monkeypatch.setenv("MI355CV_PYR_FUSE", "0")
            ref = cv.buildPyramidBatch(d, ml, border)
            monkeypatch.delenv("MI355CV_PYR_FUSE")
            for l in range(1, ml + 1):
                assert torch.equal(got[l], ref[l]), (w, h, border, l)
            lvl = fr[n - 1]
            for l in range(1, ml + 1):
                lvl = orc.orc_pyrDown(lvl, None, border)
                assert np.array_equal(got[l][n - 1].cpu().numpy(), lvl), (w, h, border, l)
    # level arrays that are views into wider parents (odd byte offsets are declined to the level-by-level path, even ones are served)
    fr = rng.integers(0, 256, (2, 300, 420), dtype=np.uint8)
    d = dev(fr)
    shapes = [(150, 210), (75, 105), (38, 53), (19, 27)]
    for off in (2, 1):
        parents = [torch.zeros((2, hh + 3, ww + 8), dtype=torch.uint8, device="cuda") for hh, ww in shapes]
        dst = [d] + [p[:, 1:1 + hh, off:off + ww] for p, (hh, ww) in zip(parents, shapes)]
        got = cv.buildPyramidBatch(d, 4, dst=dst)
        lvl = fr[1]
        for l in range(1, 5):
            lvl = orc.orc_pyrDown(lvl)
            assert np.array_equal(got[l][1].cpu().numpy(), lvl), (off, l)
            assert int(parents[l - 1][:, 0].sum()) == 0 and int(parents[l - 1][:, :, :off].sum()) == 0


def test_build_pyramid_config4(cv, orc):
    """BASELINE config 4: cornerHarris(2,3,0.04) + buildPyramid(maxlevel=4) on 1920x1080 CV_8UC1 frames (batched)."""
    rng = np.random.default_rng(809564)
    frames = rng.integers(0, 256, (3, 1080, 1920), dtype=np.uint8)
    d = dev(frames)
    pyr = cv.buildPyramidBatch(d, 4)
    assert [tuple(p.shape[1:]) for p in pyr] == [(1080, 1920), (540, 960), (270, 480), (135, 240), (68, 120)]
    single = cv.buildPyramid(d[1], 4)
    lvl = frames[1]
    for l in range(1, 5):
        lvl = orc.orc_pyrDown(lvl)
        assert np.array_equal(pyr[l][1].cpu().numpy(), lvl)
        assert np.array_equal(single[l].cpu().numpy(), lvl)
    resp = cv.cornerHarrisBatch(d, 2, 3, 0.04)
    want = orc.orc_cornerHarris(frames[2], 2, 3, 0.04)
    assert orc.rel_err(resp[2].cpu().numpy(), want) <= 1e-4


def test_good_features_to_track(cv, orc):
    for (w, h) in [(160, 120), (97, 143), (640, 480)]:
        img = structured(h, w, 11 + w)
        for harris in (False, True):
            for maxc, q, md in [(50, 0.01, 5.0), (0, 0.05, 0.0), (25, 0.02, 12.3), (1000, 0.001, 1.0)]:
                want = orc.orc_goodFeaturesToTrack(img, maxc, q, md, None, 3, 3, harris, 0.04)
                got = cv.goodFeaturesToTrack(dev(img), maxc, q, md, None, 3, 3, harris, 0.04)
                assert len(want) > 0
                # the response feeding the ranking agrees to ~1e-6 relative, not bit for bit: allow re-ordering only
                # among corners whose responses are within that noise (in practice the lists are identical)
                assert got.shape == want.shape
                assert np.array_equal(got, want) or set(map(tuple, got)) == set(map(tuple, want)), (w, h, harris, maxc, q, md)
    img = structured(120, 160, 5)
    mask = np.zeros((120, 160), np.uint8); mask[20:100, 30:120] = 255
    want = orc.orc_goodFeaturesToTrack(img, 40, 0.01, 4.0, mask, 3, 3, False, 0.04)
    got = cv.goodFeaturesToTrack(dev(img), 40, 0.01, 4.0, dev(mask))
    assert got.shape == want.shape and set(map(tuple, got)) == set(map(tuple, want))


def test_good_features_near_ties(cv, orc):
    """A periodic scene: every corner of the pattern has the same neighbourhood, so the reference's responses tie exactly (its sort then orders ties by
    address) while responses that agree to 1e-6 relative need not.  What must hold: with nothing truncated the SETS of corners are equal; where the
    list is truncated (maxCorners) or thinned (minDistance) the two lists may differ only among corners whose responses tie within that noise."""
    yy, xx = np.mgrid[0:240, 0:320]
    img = (((xx // 16 + yy // 16) % 2) * 140 + 40 + ((xx // 16) % 3) * 9).astype(np.uint8)          # checkerboard, three column families of equal contrast steps
    for harris in (False, True):
        want = orc.orc_goodFeaturesToTrack(img, 0, 0.01, 0.0, None, 3, 3, harris, 0.04)
        got, q = cv.goodFeaturesToTrack(dev(img), 0, 0.01, 0.0, None, 3, 3, harris, 0.04, returnQuality=True)
        assert len(want) > 100 and set(map(tuple, got)) == set(map(tuple, want)), harris
        qual = {tuple(c): v for c, v in zip(got, q)}
        for maxc, md in [(40, 0.0), (0, 20.0), (25, 9.0)]:
            w2 = orc.orc_goodFeaturesToTrack(img, maxc, 0.01, md, None, 3, 3, harris, 0.04)
            g2 = cv.goodFeaturesToTrack(dev(img), maxc, 0.01, md, None, 3, 3, harris, 0.04)
            assert len(g2) == len(w2), (harris, maxc, md)
            diff = set(map(tuple, g2)) ^ set(map(tuple, w2))
            if diff:                                                        # only corners that tie with one another may be exchanged
                vals = np.array([qual[c] for c in diff])
                assert vals.max() - vals.min() <= 2e-5 * abs(vals.max()), (harris, maxc, md, len(diff))
