"""GPU parity for the CV_8U YUV family through cv_hal_cvtBGRtoYUV / cvtYUVtoBGR / cvtTwoPlaneYUVtoBGR:
every code, 3/4-channel sources and destinations, odd and large widths, host pointers; bit-exact against the oracle."""
import numpy as np
import pytest
import torch

import orc as O
from test_oracle_yuv import _img

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def test_yuv_family(cv, orc):
    for (w, h) in [(1, 1), (7, 3), (64, 5), (263, 9), (1030, 2), (1920, 16)]:
        for code in list(O._YUV_FWD) + list(O._HSV):             # YUV / YCrCb and HSV (180 and 256 hue ranges)
            for cn in (3, 4):
                src = _img(h, w, cn, code + cn)
                assert np.array_equal(cv.cvtColor(torch.from_numpy(src).cuda(), code).cpu().numpy(), orc.orc_cvtColorYUV(src, code)), (w, h, code, cn)
        for code in O._YUV_INV:
            src = _img(h, w, 3, code)
            assert np.array_equal(cv.cvtColor(torch.from_numpy(src).cuda(), code).cpu().numpy(), orc.orc_cvtColorYUV(src, code)), (w, h, code)
            got4 = cv.cvtColor(torch.from_numpy(src).cuda(), code, dstCn=4).cpu().numpy()
            assert np.array_equal(got4[..., :3], orc.orc_cvtColorYUV(src, code)) and (got4[..., 3] == 255).all()
    src = _img(30, 50, 3, 1)
    assert np.array_equal(cv.cvtColor(src, cv.COLOR_BGR2YCrCb), orc.orc_cvtColorYUV(src, 36))                     # host pointers


def test_nv12_nv21(cv, orc):
    for (w, h) in [(2, 2), (6, 4), (64, 8), (262, 6), (1030, 4), (1920, 1080), (642, 362)]:
        rng = np.random.default_rng(w)
        src = rng.integers(0, 256, (h * 3 // 2, w), dtype=np.uint8)
        for code in list(O._YUV_NV) + list(O._YUV_3P):          # NV12 / NV21 and I420 / YV12 (h % 4 == 2 shifts the second chroma plane)
            got = cv.cvtColor(torch.from_numpy(src).cuda(), code).cpu().numpy()
            assert np.array_equal(got, orc.orc_cvtColorYUV(src, code)), (w, h, code)
    src = np.random.default_rng(3).integers(0, 256, (36, 40), dtype=np.uint8)
    assert np.array_equal(cv.cvtColor(src, cv.COLOR_YUV2BGR_NV12), orc.orc_cvtColorYUV(src, 91))                   # host pointers
    with pytest.raises(ValueError):
        cv.cvtColor(torch.zeros((35, 40), dtype=torch.uint8, device="cuda"), cv.COLOR_YUV2BGR_NV12)


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_yuv_wide_depths(cv, orc, dtype):
    """CV_16U (integer formulas, bit-exact) and CV_32F (<= 1e-4 relative; the forward restatement follows the reference's vector body / scalar tail
    split, the kernel uses the body's form everywhere) members of the YUV / YCrCb family, both directions, 3 and 4 channels"""
    rng = np.random.default_rng(8)
    for (w, h) in [(1, 1), (7, 3), (33, 5), (643, 48), (1920, 1080)]:
        for scn in (3, 4):
            src = rng.integers(0, 65536, (h, w, scn), dtype=np.uint16) if dtype == np.uint16 else rng.random((h, w, scn), dtype=np.float32)
            for code in (82, 83, 36, 37):
                got = cv.cvtColor(torch.from_numpy(src).cuda(), code).cpu().numpy()
                want = orc.orc_cvtColorYUVwide(src, code)
                assert (np.array_equal(got, want) if dtype == np.uint16 else orc.rel_err(got, want) <= 1e-6), (w, h, scn, code)
        src = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16) if dtype == np.uint16 else rng.random((h, w, 3), dtype=np.float32)
        for code in (84, 85, 38, 39):
            for dcn in (3, 4):
                got = cv.cvtColor(torch.from_numpy(src).cuda(), code, dstCn=dcn).cpu().numpy()
                want = orc.orc_cvtColorYUVwide(src, code, dcn)
                assert (np.array_equal(got, want) if dtype == np.uint16 else orc.rel_err(got, want) <= 1e-6), (w, h, dcn, code)
