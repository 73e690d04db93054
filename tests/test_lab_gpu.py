"""cv_hal_cvtBGRtoLab / cv_hal_cvtLabtoBGR (L*a*b* and L*u*v*, CV_8U) on the GPU (csrc/color_lab.hip) against oracle/color_lab.c, which tests/test_oracle_lab.py pins to the
reference for every 8-bit colour: again EVERY colour in both directions (sRGB and linear), then the channel orders, 4-channel forms, ragged widths,
unaligned views and host-resident images."""
import numpy as np
import pytest
import torch

import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def all_colours():
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.empty((4096, 4096, 3), np.uint8)
    img[..., 0] = (v & 255).reshape(4096, 4096)
    img[..., 1] = ((v >> 8) & 255).reshape(4096, 4096)
    img[..., 2] = (v >> 16).reshape(4096, 4096)
    return img


@pytest.mark.parametrize("code", [44, 75, 56, 79, 50, 51, 58, 81])
def test_every_8bit_colour(cv, code):
    img = all_colours()
    got = cv.cvtColor(torch.from_numpy(img).cuda(), code).cpu().numpy()
    want = orc.orc_cvtColorLab(img, code)
    bad = np.flatnonzero((want != got).any(axis=2).ravel())
    assert bad.size == 0, (code, bad.size, img.reshape(-1, 3)[bad[:5]], want.reshape(-1, 3)[bad[:5]], got.reshape(-1, 3)[bad[:5]])


@pytest.mark.parametrize("code", [44, 45, 74, 75, 50, 51])
@pytest.mark.parametrize("scn", [3, 4])
def test_forward(cv, code, scn):
    rng = np.random.default_rng(code * 10 + scn)
    for (h, w) in [(1, 1), (3, 7), (61, 333), (240, 641), (35, 1024)]:
        img = rng.integers(0, 256, (h, w, scn), dtype=np.uint8)
        want = orc.orc_cvtColorLab(img, code)
        assert np.array_equal(cv.cvtColor(torch.from_numpy(img).cuda(), code).cpu().numpy(), want), (code, scn, h, w)
        assert np.array_equal(cv.cvtColor(img, code), want), ("host", code, scn, h, w)
    parent = torch.from_numpy(rng.integers(0, 256, (70, 301, scn), dtype=np.uint8)).cuda()         # a view: odd row pitch, odd offset
    view = parent[3:64, 5:298]
    assert np.array_equal(cv.cvtColor(view, code).cpu().numpy(), orc.orc_cvtColorLab(np.ascontiguousarray(view.cpu().numpy()), code))


@pytest.mark.parametrize("code", [56, 57, 78, 79, 58, 59, 80, 81])
@pytest.mark.parametrize("dcn", [3, 4])
def test_inverse(cv, code, dcn):
    rng = np.random.default_rng(code * 10 + dcn)
    for (h, w) in [(1, 1), (3, 7), (61, 333), (240, 641), (35, 1024)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = orc.orc_cvtColorLab(img, code, dcn)
        assert np.array_equal(cv.cvtColor(torch.from_numpy(img).cuda(), code, dstCn=dcn).cpu().numpy(), want), (code, dcn, h, w)
        assert np.array_equal(cv.cvtColor(img, code, dstCn=dcn), want), ("host", code, dcn, h, w)
    parent = torch.from_numpy(rng.integers(0, 256, (70, 301, 3), dtype=np.uint8)).cuda()
    view = parent[3:64, 5:298]
    assert np.array_equal(cv.cvtColor(view, code, dstCn=dcn).cpu().numpy(), orc.orc_cvtColorLab(np.ascontiguousarray(view.cpu().numpy()), code, dcn))


def _err(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(1.0, np.abs(b))))


@pytest.mark.parametrize("code", [44, 45, 74, 75, 50, 51, 76, 77])
@pytest.mark.parametrize("scn", [3, 4])
def test_float_forward(cv, code, scn):
    """CV_32F L*a*b*: the kernels evaluate the oracle's expressions operation for operation (IEEE, no contraction): equal bits expected, north_star's
    1e-4 asserted; the oracle is pinned to the reference in tests/test_oracle_lab.py"""
    rng = np.random.default_rng(code + scn)
    for (h, w) in [(1, 1), (3, 7), (61, 333), (240, 641), (1080, 1920)]:
        img = (rng.random((h, w, scn), dtype=np.float32) * 1.2 - 0.1).astype(np.float32)
        want = orc.orc_cvtColorLab(img, code)
        got = cv.cvtColor(torch.from_numpy(img).cuda(), code).cpu().numpy()
        assert _err(got, want) <= 1e-4, (code, scn, h, w, _err(got, want))
        assert np.mean(got == want) > 0.999, (code, scn, h, w, float(np.mean(got == want)))
    assert _err(cv.cvtColor(img, code), want) <= 1e-4                                               # host-resident image


@pytest.mark.parametrize("code", [56, 57, 78, 79, 58, 59, 80, 81])
@pytest.mark.parametrize("dcn", [3, 4])
def test_float_inverse(cv, code, dcn):
    rng = np.random.default_rng(code + dcn)
    for (h, w) in [(1, 1), (3, 7), (61, 333), (240, 641), (1080, 1920)]:
        lab = np.empty((h, w, 3), np.float32)
        lab[..., 0] = rng.random((h, w)) * 100
        if code in (56, 57, 78, 79):
            lab[..., 1:] = rng.random((h, w, 2)) * 254 - 127
        else:
            lab[..., 1] = rng.random((h, w)) * 354 - 134
            lab[..., 2] = rng.random((h, w)) * 262 - 140
        want = orc.orc_cvtColorLab(lab, code, dcn)
        got = cv.cvtColor(torch.from_numpy(lab).cuda(), code, dstCn=dcn).cpu().numpy()
        assert _err(got, want) <= 1e-4, (code, dcn, h, w, _err(got, want))
        assert np.mean(got == want) > 0.999, (code, dcn, h, w, float(np.mean(got == want)))


@pytest.mark.parametrize("code", [76, 77])
@pytest.mark.parametrize("scn", [3, 4])
def test_luv_from_linear_rgb_8u(cv, code, scn):
    rng = np.random.default_rng(code + scn)
    for (h, w) in [(1, 1), (3, 7), (61, 333), (240, 641), (1080, 1920)]:
        img = rng.integers(0, 256, (h, w, scn), dtype=np.uint8)
        want = orc.orc_cvtColorLab(img, code)
        assert np.array_equal(cv.cvtColor(torch.from_numpy(img).cuda(), code).cpu().numpy(), want), (code, scn, h, w)
    if scn == 3:
        img = all_colours()
        assert np.array_equal(cv.cvtColor(torch.from_numpy(img).cuda(), code).cpu().numpy(), orc.orc_cvtColorLab(img, code)), code


def test_declines(cv):
    with pytest.raises((NotImplementedError, ValueError)):
        cv.cvtColor(torch.zeros((8, 8, 3), dtype=torch.int16, device="cuda"), cv.COLOR_BGR2Luv)
    L = cv._lib.lib
    a = torch.zeros((8, 8, 3), dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    assert L.mi355cv_cvtBGRtoLab(a.data_ptr(), 24, b.data_ptr(), 24, 4, 8, 2, 3, False, False, False) == 1         # CV_16U: neither path of the reference's hook pair
    assert L.mi355cv_cvtLabtoBGR(a.data_ptr(), 24, b.data_ptr(), 24, 8, 8, 0, 2, False, True, True) == 1           # two destination channels
