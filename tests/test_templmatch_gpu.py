"""GPU parity for row a13: cv::matchTemplate (all six methods; direct and MFMA-i8 paths) and cv::integral -- through the
C ABI against the oracle.  Results are CV_32F: 1e-4 relative (reference's own bound vs brute force is 1e-3)."""
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


def rnd(shape, dtype, seed):
    rng = np.random.default_rng(seed)
    return rng.random(shape, dtype=np.float32) if dtype == np.float32 else rng.integers(0, 256, shape, dtype=np.uint8)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("cn", [1, 3])
@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5])
def test_modes_direct_path(cv, orc, dtype, cn, method):
    for (iw, ih, tw, th) in [(64, 48, 8, 8), (97, 61, 17, 9), (40, 40, 40, 40), (130, 33, 5, 30)]:
        img = rnd((ih, iw, cn) if cn > 1 else (ih, iw), dtype, 100 + iw)
        tpl = rnd((th, tw, cn) if cn > 1 else (th, tw), dtype, 200 + tw)
        want = orc.orc_matchTemplate(img, tpl, method)
        got = cv.matchTemplate(dev(img), dev(tpl), method).cpu().numpy()
        # CV_32FC1 with >= 4096 results runs as three bf16 products on the matrix cores (test_bf16_split_path_32fc1): the contract's 1e-4 there, 1e-5 on the direct kernel
        from opencv_amd import _lib
        tol = 1e-4 if "k_ccorr_bf16" in _lib.lib.mi355cv_lastKernel().decode() else 1e-5
        assert orc.rel_err(got, want) <= tol, (iw, ih, tw, th)
    img, tpl = rnd((61, 97), dtype, 1), rnd((9, 17), dtype, 2)
    got = cv.matchTemplate(img, tpl, method)                                                                    # host arrays
    tol = 1e-4 if "k_ccorr_bf16" in _lib.lib.mi355cv_lastKernel().decode() else 1e-5
    assert orc.rel_err(got, orc.orc_matchTemplate(img, tpl, method)) <= tol


@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5])
def test_mfma_path_8uc1(cv, orc, method):
    """>= 4096 outputs, CV_8UC1, template <= 128x128 -> v_mfma_i32_32x32x32_i8 path; the integer correlation is exact"""
    for (iw, ih, tw, th) in [(300, 200, 16, 16), (513, 301, 128, 128), (400, 390, 33, 77), (700, 150, 100, 5), (1000, 130, 128, 1)]:
        img = rnd((ih, iw), np.uint8, 10 + iw)
        tpl = rnd((th, tw), np.uint8, 20 + tw)
        want = orc.orc_matchTemplate(img, tpl, method)
        got = cv.matchTemplate(dev(img), dev(tpl), method).cpu().numpy()
        if method == 2:
            assert np.array_equal(got, want), (iw, ih, tw, th)          # exact integers rounded to float once
        else:
            assert orc.rel_err(got, want) <= 1e-6, (iw, ih, tw, th)


@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5])
def test_templates_beyond_128_as_blocks(cv, orc, method):
    """CV_8UC1 templates of 129 .. 512 per side: up to 4 x 4 blocks of <= 128 x 128 on the matrix cores (the correlation is linear in the template; exact int32 planes summed
    as integers) -- the same exactness as the single-block path: TM_CCORR identical to the restatement, the normalised methods within 1e-6.  Batches go the same way."""
    from opencv_amd import _lib
    for (iw, ih, tw, th) in [(400, 330, 129, 129), (520, 300, 200, 90), (300, 420, 60, 256), (600, 400, 256, 256), (450, 350, 131, 255), (640, 600, 300, 385), (700, 580, 512, 130)]:
        if tw * th > 70000 and method not in (0, 5):
            continue                                                            # (the 9- and 10-block cases on two methods: the restatement is the slow side)
        img = rnd((ih, iw), np.uint8, 30 + iw)
        tpl = rnd((th, tw), np.uint8, 40 + tw)
        if method in (1, 3, 5):
            img[10:10 + th, 20:20 + tw] = tpl                                   # a perfect match inside: the normalised scores reach their extreme there
        want = orc.orc_matchTemplate(img, tpl, method)
        got = cv.matchTemplate(dev(img), dev(tpl), method).cpu().numpy()
        assert "blocks of <= 128 x 128" in _lib.lib.mi355cv_lastKernel().decode(), _lib.lib.mi355cv_lastKernel().decode()
        if method == 2:
            assert np.array_equal(got, want), (iw, ih, tw, th)
        else:
            assert orc.rel_err(got, want) <= 1e-6, (iw, ih, tw, th)
    frames = rnd((3, 300, 400), np.uint8, 5)
    tpl = rnd((150, 140), np.uint8, 6)
    got = cv.matchTemplateBatch(dev(frames), dev(tpl), method).cpu().numpy()
    for f in range(3):
        want = orc.orc_matchTemplate(frames[f], tpl, method)
        assert (np.array_equal(got[f], want) if method == 2 else orc.rel_err(got[f], want) <= 1e-6), f


@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5])
def test_float_templates_beyond_128_as_blocks(cv, orc, method):
    """CV_32FC1 templates of 129 .. 512 per side: the split-bf16 products of up to 4 x 4 blocks accumulated onto one result (the path's 1e-4 bar; measured ~1e-6)"""
    from opencv_amd import _lib
    for (iw, ih, tw, th) in [(400, 330, 129, 129), (520, 300, 200, 90), (300, 420, 60, 256), (560, 400, 300, 257)]:
        if tw * th > 70000 and method not in (0, 5):
            continue
        img = rnd((ih, iw), np.float32, 50 + iw)
        tpl = rnd((th, tw), np.float32, 60 + tw)
        if method in (1, 3, 5):
            img[10:10 + th, 20:20 + tw] = tpl
        want = orc.orc_matchTemplate(img, tpl, method)
        got = cv.matchTemplate(dev(img), dev(tpl), method).cpu().numpy()
        assert "k_ccorr_bf16" in _lib.lib.mi355cv_lastKernel().decode() and "block" in _lib.lib.mi355cv_lastKernel().decode(), _lib.lib.mi355cv_lastKernel().decode()
        if method in (1, 3, 5):
            assert np.max(np.abs(got - want)) <= 1e-4, (iw, ih, tw, th, float(np.max(np.abs(got - want))))                  # normalised results live in [-1, 1]
        else:                                                                   # un-normalised: relative to |I| |T|, the scale the products' rounding acts on (as for one block)
            scale = float(np.sqrt((img.astype(np.float64) ** 2).sum() / img.size * tw * th) * np.sqrt((tpl.astype(np.float64) ** 2).sum()))
            assert np.max(np.abs(got.astype(np.float64) - want)) <= 1e-4 * scale, (iw, ih, tw, th, float(np.max(np.abs(got.astype(np.float64) - want)) / scale))


@pytest.mark.parametrize("method", [1, 3, 4, 5])
def test_mfma_fused_window_sums(cv, orc, method):
    """templates of >= 66 rows on 4-byte aligned rows: the window sums of I and I^2 come out of the MFMA kernel itself (running column
    sums + a lane prefix scan per output row); several row blocks, widths that leave partial tiles, template widths that are not
    multiples of 4, and the smallest / an odd template height"""
    for (iw, ih, tw, th) in [(516, 700, 128, 128), (640, 480, 100, 66), (772, 400, 57, 127), (260, 330, 128, 67)]:
        img = rnd((ih, iw), np.uint8, 30 + iw)
        tpl = rnd((th, tw), np.uint8, 40 + tw)
        want = orc.orc_matchTemplate(img, tpl, method)
        got = cv.matchTemplate(dev(img), dev(tpl), method).cpu().numpy()
        assert orc.rel_err(got, want) <= 1e-6, (iw, ih, tw, th)
    # the same frames through the batch entry (two workgroups of different frames per CU)
    frames = np.stack([rnd((300, 640), np.uint8, 50 + k) for k in range(5)])
    tpl = rnd((96, 80), np.uint8, 60)
    rb = cv.matchTemplateBatch(dev(frames), dev(tpl), method).cpu().numpy()
    for k in range(5):
        assert orc.rel_err(rb[k], orc.orc_matchTemplate(frames[k], tpl, method)) <= 1e-6, k


@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("cn", [3, 4, 2])
def test_mfma_path_multichannel_8u(cv, orc, cn, method, monkeypatch):
    """CV_8UC2 / C3 / C4 with >= 4096 outputs: per-channel planes through the i8 MFMA path, the exact per-channel correlations summed in
    double, the multi-channel normalisation of common_matchTemplate on top; against the oracle and against the direct kernel
    (MI355CV_TM_PLANES=0); single images (plane widths that are / are not multiples of 4) and a batch"""
    sizes = [(300, 200, 16, 16), (401, 390, 33, 77)] + ([(516, 301, 128, 128)] if method in (3, 4) else [])      # (the CPU oracle takes seconds per 128x128 case)
    for (iw, ih, tw, th) in sizes:
        img = rnd((ih, iw, cn), np.uint8, 70 + iw + cn)
        tpl = rnd((th, tw, cn), np.uint8, 80 + tw)
        want = orc.orc_matchTemplate(img, tpl, method)
        got = cv.matchTemplate(dev(img), dev(tpl), method)
        assert orc.rel_err(got.cpu().numpy(), want) <= 1e-6, (iw, ih, tw, th)
        if tw == 16:                                                   # the direct kernel takes seconds on the larger cases
            monkeypatch.setenv("MI355CV_TM_PLANES", "0")
            direct = cv.matchTemplate(dev(img), dev(tpl), method)
            monkeypatch.delenv("MI355CV_TM_PLANES")
            assert orc.rel_err(got.cpu().numpy(), direct.cpu().numpy()) <= 1e-5
    if method in (3, 5):
        frames = np.stack([rnd((260, 520, cn), np.uint8, 90 + k) for k in range(3)])
        tpl = rnd((96, 80, cn), np.uint8, 91)
        rb = cv.matchTemplateBatch(dev(frames), dev(tpl), method).cpu().numpy()
        for k in range(3):
            assert orc.rel_err(rb[k], orc.orc_matchTemplate(frames[k], tpl, method)) <= 1e-6, k


def test_template_found_and_batch(cv, orc):
    img = rnd((480, 640), np.uint8, 7)
    tpl = np.ascontiguousarray(img[100:228, 200:328])
    r = cv.matchTemplate(dev(img), dev(tpl), cv.TM_CCORR_NORMED).cpu().numpy()
    assert np.unravel_index(np.argmax(r), r.shape) == (100, 200) and abs(r[100, 200] - 1.0) <= 1e-6
    frames = np.stack([img, np.roll(img, 17, axis=1), np.roll(img, 5, axis=0)])
    rb = cv.matchTemplateBatch(dev(frames), dev(tpl), cv.TM_CCORR_NORMED).cpu().numpy()
    assert np.array_equal(rb[0], r)
    for f in (1, 2):
        assert orc.rel_err(rb[f], orc.orc_matchTemplate(frames[f], tpl, 3)) <= 1e-6


def test_config5_4k(cv, orc):
    """BASELINE config 5: TM_CCORR_NORMED, 3840x2160 CV_8UC1 x 128x128 -> 3713x2033 CV_32F (oracle on crops)."""
    img = rnd((2160, 3840), np.uint8, 809564)
    tpl = rnd((128, 128), np.uint8, 1)
    r = cv.matchTemplate(dev(img), dev(tpl), cv.TM_CCORR_NORMED)
    assert tuple(r.shape) == (2033, 3713)
    for (y0, x0) in [(0, 0), (1900, 3500), (1000, 2000)]:
        crop = np.ascontiguousarray(img[y0:y0 + 128 + 40, x0:x0 + 128 + 60])
        want = orc.orc_matchTemplate(crop, tpl, 3)
        got = r[y0:y0 + 41, x0:x0 + 61].cpu().numpy()
        assert orc.rel_err(got, want) <= 1e-6


def test_integral(cv, orc):
    """cv::integral: default depths (CV_32S sum for 8-bit sources), CV_64F sums, squared sums; sizes spanning one and several
    column segments; exact for 8-bit sources, against numpy's double cumsum to 1e-13 for float sources (bit for bit against the reference's order:
    test_integral_every_triple_in_the_reference_order)."""
    for dtype in (np.uint8, np.float32):
        for cn in (1, 3):
            for (h, w) in [(37, 53), (1, 1), (64, 300), (65, 17), (200, 129), (1080, 1920)]:
                if h * w > 10000 and cn == 3:
                    continue
                src = rnd((h, w, cn) if cn > 1 else (h, w), dtype, 3)
                a = src.astype(np.float64)
                ws = np.zeros((h + 1, w + 1) + a.shape[2:]); wq = np.zeros_like(ws)
                ws[1:, 1:] = a.cumsum(0).cumsum(1); wq[1:, 1:] = (a * a).cumsum(0).cumsum(1)
                s, q = cv.integral(dev(src), sqsum=True)
                s, q = s.cpu().numpy(), q.cpu().numpy()
                if dtype == np.uint8:
                    assert s.dtype == np.int32 and np.array_equal(s, ws.astype(np.int64)) and np.array_equal(q, wq), (h, w, cn)
                    s64 = cv.integral(dev(src), sdepth=6).cpu().numpy()
                    assert s64.dtype == np.float64 and np.array_equal(s64, ws)
                else:
                    assert s.dtype == np.float64 and np.allclose(s, ws, rtol=1e-13, atol=0) and np.allclose(q, wq, rtol=1e-13, atol=0)
    src = rnd((40, 70), np.uint8, 8)
    s = cv.integral(src)                                                 # host pointers
    assert isinstance(s, np.ndarray) and s[-1, -1] == int(src.sum())


def test_integral_row_phases(cv, orc):
    """CV_32S sums store whole 16-byte aligned pieces whatever the phase of the W + 1-int rows (integral.hip storeRowAligned): every width class mod 4 around the
    256-column tile boundaries, single frames and batches (whose frames start at every phase too), exact"""
    from opencv_amd import _lib
    for w in (252, 253, 254, 255, 256, 257, 258, 259, 260, 509, 510, 511, 512, 513, 767, 1023, 1025, 3, 4, 5):
        for h in (1, 2, 5, 17, 33):
            src = rnd((h, w), np.uint8, w + h)
            ws = np.zeros((h + 1, w + 1), np.int64); ws[1:, 1:] = src.astype(np.int64).cumsum(0).cumsum(1)
            got = cv.integral(dev(src)).cpu().numpy()
            assert got.dtype == np.int32 and np.array_equal(got, ws), (w, h)
    fr = rnd((5, 33, 258), np.uint8, 9)
    out = cv.integralBatch(dev(fr)).cpu().numpy()
    for f in range(5):
        ws = np.zeros((34, 259), np.int64); ws[1:, 1:] = fr[f].astype(np.int64).cumsum(0).cumsum(1)
        assert np.array_equal(out[f], ws), f


@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5])
def test_bf16_split_path_32fc1(cv, orc, method):
    """CV_32FC1 with >= 4096 outputs and a template <= 128x128: three bf16 products (hi*hi + hi*mid + mid*hi) on v_mfma_f32_32x32x16_bf16, fp32
    accumulation -- within 1e-4 of the float64 restatement (the contract; measured ~1e-6 on non-negative data), every method, template sizes on both
    sides of the 16-column K steps and of the 32-row template chunks, widths that leave partial tiles, zero-mean data"""
    from opencv_amd import _lib
    for (iw, ih, tw, th) in [(300, 200, 16, 16), (513, 301, 128, 128), (400, 390, 33, 77), (700, 150, 100, 5), (1000, 130, 128, 1), (260, 330, 47, 97), (263, 200, 1, 1)]:
        img = rnd((ih, iw), np.float32, 10 + iw)
        tpl = rnd((th, tw), np.float32, 20 + tw)
        for shift in (0.0, 0.5):                                          # [0, 1) and zero-mean [-0.5, 0.5)
            if shift and tw * th < 64:
                continue                                                  # a normalised 1x1 "window" of zero-mean data is +-1 with a vanishing denominator: no tolerance is meaningful
            a, b = (img - shift).astype(np.float32), (tpl - shift).astype(np.float32)
            want = orc.orc_matchTemplate(a, b, method)
            got = cv.matchTemplate(dev(a), dev(b), method).cpu().numpy()
            assert "k_ccorr_bf16" in _lib.lib.mi355cv_lastKernel().decode(), _lib.lib.mi355cv_lastKernel().decode()
            if method in (1, 3, 5):
                assert np.max(np.abs(got - want)) <= 1e-4, (iw, ih, tw, th, shift, float(np.max(np.abs(got - want))))      # normalised results live in [-1, 1]
            else:
                # un-normalised: relative to |I| |T|, the scale the products' rounding acts on
                scale = float(np.sqrt((a.astype(np.float64) ** 2).sum() / a.size * tw * th) * np.sqrt((b.astype(np.float64) ** 2).sum()))
                assert np.max(np.abs(got.astype(np.float64) - want)) <= 1e-4 * max(scale, 1e-12), (iw, ih, tw, th, shift)
    # batches go through the same kernels (grid z = frame)
    fr = rnd((3, 200, 300), np.float32, 5); tpl = rnd((31, 45), np.float32, 6)
    got = cv.matchTemplateBatch(dev(fr), dev(tpl), 3).cpu().numpy()
    for f in range(3):
        assert orc.rel_err(got[f], orc.orc_matchTemplate(fr[f], tpl, 3)) <= 1e-4, f


def test_bf16_split_path_on_offset_images_and_non_finite_pixels(cv, orc):
    """TM_SQDIFF* / TM_CCOEFF* are differences of large terms, so the bf16 split's dropped mid * mid product (2^-17 relative) is amplified near a perfect match
    and on images with a large offset -- those methods take the fourth product.  A high-DC float image that contains the template exactly: the match location is found and
    the result stays within 1e-4 of the float64 restatement's scale; an Inf pixel gives Inf / NaN only in the windows that contain it, as the fp32 form does."""
    from opencv_amd import _lib
    rng = np.random.default_rng(77)
    img = (1000.0 + rng.random((200, 320), dtype=np.float32) * 8).astype(np.float32)             # offset 1000, signal 8
    tpl = np.ascontiguousarray(img[60:60 + 48, 100:100 + 64])
    for method in (0, 1, 4, 5):
        want = orc.orc_matchTemplate(img, tpl, method)
        got = cv.matchTemplate(dev(img), dev(tpl), method).cpu().numpy()
        k = _lib.lib.mi355cv_lastKernel().decode()
        assert "k_ccorr_bf16" in k and "mid*mid" in k, k
        loc = np.unravel_index(np.argmin(got) if method < 2 else np.argmax(got), got.shape)
        assert loc == (60, 100), (method, loc)
        if method in (1, 5):
            # normalised results divide two differences of terms near 3e9 whose difference is ~1e4: no fp32 accumulation (the reference's float FFT included) keeps
            # four digits of the field there, so the field is not compared -- the match itself must stand out
            assert (got[60, 100] <= 1e-4 and np.partition(got.ravel(), 1)[1] > 10 * max(float(got[60, 100]), 1e-7)) if method == 1 else got[60, 100] >= 0.95, (method, float(got[60, 100]))
        else:
            scale = float((img.astype(np.float64) ** 2).mean() * tpl.size)
            assert np.max(np.abs(got.astype(np.float64) - want)) <= 1e-4 * scale, method
    a = rnd((150, 300), np.float32, 31); a[70, 140] = np.inf
    t = rnd((20, 30), np.float32, 32)
    got = cv.matchTemplate(dev(a), dev(t), 2).cpu().numpy()                                                       # TM_CCORR
    hit = np.zeros(got.shape, bool); hit[max(0, 70 - 19):71, max(0, 140 - 29):141] = True                        # the windows that contain the pixel
    assert not np.isfinite(got[hit]).any()
    # the Toeplitz GEMM also multiplies the pixel with the ZERO taps that pad a template row to its K steps (Inf * 0 = NaN), so the non-finite patch is up to
    # 31 columns wider on either side than the reference's; beyond that nothing is touched
    far = np.ones(got.shape, bool); far[max(0, 70 - 19):71, max(0, 140 - 29 - 32):141 + 32] = False
    assert np.isfinite(got[far]).all()


def _mask(th, tw, cn, kind, seed):
    rng = np.random.default_rng(seed)
    shape = (th, tw, cn) if (kind.endswith("cn") and cn > 1) else (th, tw)
    if kind.startswith("u8"):
        return np.array([0, 1, 7, 255], np.uint8)[rng.integers(0, 4, shape)]
    return rng.random(shape, dtype=np.float32)


def _mask_tol(img, tpl, mask, method, got, want):
    """normalised methods live in [-1, 1]: absolute 1e-4; the others relative to |I| |T M| (the scale the products' rounding acts on)"""
    if method in (1, 3, 5):
        return float(np.max(np.abs(got - want))), 1e-4
    m = mask.astype(np.float64) if mask.dtype == np.float32 else (mask > 0).astype(np.float64)
    if m.ndim < tpl.ndim:
        m = m[..., None]
    scale = float(np.sqrt((img.astype(np.float64) ** 2).mean() * img.shape[-1] ** (img.ndim == 3) * tpl.shape[0] * tpl.shape[1]) * np.sqrt(((tpl.astype(np.float64) * m) ** 2).sum()))
    if method in (0, 4):
        scale = max(scale, float((img.astype(np.float64) ** 2).mean()) * tpl.size)
    return float(np.max(np.abs(got.astype(np.float64) - want))), 1e-4 * max(scale, 1e-12)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("cn,kind", [(1, "u8"), (1, "f32"), (3, "u8"), (3, "u8cn"), (3, "f32"), (3, "f32cn")])     # (a per-channel mask of a one-channel template is the one-channel mask)
def test_masked_modes(cv, orc, dtype, cn, kind):
    """matchTemplateMask (templmatch.cpp:762-904), every method, CV_8U and CV_32F images, binary and weighted masks with one channel or the template's: the small
    shapes run the direct kernel, the large ones the bf16 matrix-core correlations (four partial products) -- CV_8U images under a binary mask the i8 ones for the
    methods without a mean --; against the restatement pinned to the reference"""
    from opencv_amd import _lib
    for (iw, ih, tw, th) in [(64, 48, 8, 8), (97, 61, 17, 9), (300, 200, 33, 21), (261, 190, 128, 64)]:
        img = rnd((ih, iw, cn) if cn > 1 else (ih, iw), dtype, 300 + iw)
        tpl = rnd((th, tw, cn) if cn > 1 else (th, tw), dtype, 400 + tw)
        mask = _mask(th, tw, cn, kind, 500 + tw)
        for method in range(6):
            want = orc.orc_matchTemplateMask(img, tpl, method, mask)
            got = cv.matchTemplate(dev(img), dev(tpl), method, mask=dev(mask)).cpu().numpy()
            last = _lib.lib.mi355cv_lastKernel().decode()
            if dtype == np.uint8 and kind.startswith("u8") and method <= 3:
                assert "byte planes" in last, last                 # CV_8U under a binary mask: exact integer correlations on the i8 matrix-core path
            elif iw >= 261:
                assert "k_ccorr_bf16" in last and "mid*mid" in last, last
            err, tol = _mask_tol(img, tpl, mask, method, got, want)
            assert np.isfinite(got).all() and err <= tol, (iw, ih, tw, th, method, err, tol)
    # host arrays in, host array out; a result row pitch that is not the width
    img, tpl, mask = rnd((61, 97), dtype, 1), rnd((9, 17), dtype, 2), _mask(9, 17, 1, "u8", 3)
    got = cv.matchTemplate(img, tpl, 5, mask=mask)
    assert np.max(np.abs(got - orc.orc_matchTemplateMask(img, tpl, 5, mask))) <= 1e-4


def test_masked_match_is_found(cv, orc):
    """a template cut out of a 1080p frame with half of it masked away and the masked half of the template overwritten: every method puts its extremum there, the
    normalised scores are 1 / 0 there; an all-ones mask gives the unmasked TM_CCORR / TM_SQDIFF results"""
    img = rnd((1080, 1920), np.uint8, 9)
    tpl = np.ascontiguousarray(img[400:400 + 96, 700:700 + 128]).copy()
    mask = np.zeros(tpl.shape, np.uint8); mask[:, :64] = 255
    tpl[:, 64:] = 99
    for method, val in [(0, 0.0), (1, 0.0), (2, None), (3, 1.0), (4, None), (5, 1.0)]:
        got = cv.matchTemplate(dev(img), dev(tpl), method, mask=dev(mask)).cpu().numpy()
        if method != 2:                                                                     # (TM_CCORR without normalisation peaks where the image is bright)
            loc = np.unravel_index(np.argmin(got) if method < 2 else np.argmax(got), got.shape)
            assert loc == (400, 700), (method, loc)
        if val is not None:
            # TM_SQDIFF is a difference of terms near |T M|^2 (no clamp at zero in matchTemplateMask, :811): zero to fp32 rounding of those terms
            tol = 1e-4 if method != 0 else 1e-6 * float((tpl[:, :64].astype(np.float64) ** 2).sum())
            assert abs(float(got[400, 700]) - val) <= tol, (method, float(got[400, 700]))
    ones = np.ones((96, 128), np.uint8)
    tpl2 = np.ascontiguousarray(img[400:400 + 96, 700:700 + 128])
    for method in (0, 2):
        a = cv.matchTemplate(dev(img), dev(tpl2), method, mask=dev(ones)).cpu().numpy()
        b = cv.matchTemplate(dev(img), dev(tpl2), method).cpu().numpy()
        # the unmasked path sums exact byte products in integers (i8 matrix cores); the masked one works on float planes and accumulates 12 288 products in
        # fp32 (ulp 16-32 at 1.3e8-2.7e8; measured: 20 ulp), and TM_SQDIFF is |I M|^2 - 2 (I M).(T M) + |T M|^2 of three such terms (templmatch.cpp:807-811)
        assert np.max(np.abs(a - b)) <= 3e-5 * float(np.max(np.abs(b))), method


def test_masked_argument_checks(cv):
    """the reference's assertions (templmatch.cpp:764-767) are ValueErrors in the mirror; the C ABI declines what it is not given properly, with a reason"""
    from opencv_amd import _lib
    img, tpl = rnd((64, 80), np.uint8, 1), rnd((8, 8), np.uint8, 2)
    with pytest.raises(ValueError):
        cv.matchTemplate(dev(img), dev(tpl), 3, mask=dev(np.ones((8, 9), np.uint8)))
    with pytest.raises(ValueError):
        cv.matchTemplate(dev(img), dev(tpl), 3, mask=dev(np.ones((8, 8), np.int16)))
    d = dev(img); t = dev(tpl); r = torch.empty((57, 73), dtype=torch.float32, device="cuda")
    rc = _lib.lib.mi355cv_matchTemplateMask(d.data_ptr(), 80, 80, 64, t.data_ptr(), 8, 8, 8, 0, None, 8, 0, r.data_ptr(), 73 * 4, 3)
    assert rc == 1 and b"mask" in _lib.lib.mi355cv_lastError()


ORDERED_TRIPLES = [(np.uint8, 4, 6), (np.uint8, 4, 5), (np.uint8, 4, 4), (np.uint8, 5, 6), (np.uint8, 5, 5), (np.uint8, 6, 6), (np.uint16, 6, 6), (np.int16, 6, 6),
                   (np.float32, 5, 6), (np.float32, 5, 5), (np.float32, 6, 6), (np.float64, 6, 6)]


def _bits(a):
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize]) if a.dtype.kind == "f" else a


@pytest.mark.parametrize("dtype,sdepth,sqdepth", ORDERED_TRIPLES)
def test_integral_every_triple_in_the_reference_order(cv, orc, dtype, sdepth, sqdepth):
    """cv::integral for every row of the reference's type table, with and without the squared and the tilted sum: float sums, CV_32F / CV_32S squared sums and tilted
    sums are accumulated in the reference's order (integral_seq.hip), so they equal the restatement (tests/test_oracle_integral.py: == cv::integral) in every BIT;
    sizes around the kernels' chunking (64-row column chunks, 256-thread diagonal groups), widths 1 / 2, 1-4 channels."""
    from test_oracle_integral import source
    for (h, w) in [(1, 1), (1, 9), (9, 1), (5, 2), (3, 3), (66, 130), (130, 66), (300, 517)]:
        for cn in (1, 2, 3, 4):
            if cn in (2, 4) and h * w > 1000:
                continue
            for (sq, tl) in ((False, False), (True, False), (False, True), (True, True)):
                if sqdepth != 6 and not sq:
                    continue
                src = source((h, w, cn) if cn > 1 else (h, w), dtype, 7 * h + w + cn)
                want = orc.orc_integral(src, sdepth, sqdepth, sq, tl)
                if dtype == np.uint8 and sdepth == 5 and not sq and not tl and h * w * 255 >= 2 ** 24:
                    with pytest.raises(NotImplementedError):                              # the reference's bits depend on the CPU's vector width there: declined
                        cv.integral(torch.from_numpy(src).cuda(), sdepth=sdepth, sqdepth=sqdepth)
                    continue
                got = cv.integral(torch.from_numpy(src).cuda(), sqsum=sq, sdepth=sdepth, sqdepth=sqdepth, tilted=tl)
                got = list(got) if isinstance(got, tuple) else [got]
                for g, r, name in zip(got, [x for x in want if x is not None], [n for n, x in zip(("sum", "sqsum", "tilted"), want) if x is not None]):
                    g = g.cpu().numpy()
                    assert g.dtype == r.dtype and np.array_equal(_bits(g), _bits(r)), (name, h, w, cn, sq, tl, np.argwhere(_bits(g) != _bits(r))[:4])


def test_integral_ordered_full_frame_and_host_images(cv, orc):
    """a 1080p CV_32F frame with all three outputs (bit for bit), host pointers, wrapping CV_32S sums, and the one declined case"""
    from test_oracle_integral import source
    src = source((1080, 1920), np.float32, 5)
    want = orc.orc_integral(src, 5, 6, True, True)
    got = cv.integral(dev(src), sqsum=True, sdepth=5, tilted=True)
    for g, r in zip(got, want):
        assert np.array_equal(_bits(g.cpu().numpy()), _bits(r))
    from opencv_amd import _lib
    assert "k_iseq" in _lib.lib.mi355cv_lastKernel().decode()
    small = source((40, 70, 3), np.float32, 6)
    s, t = cv.integral(small, sdepth=5, tilted=True)                       # host pointers
    w = orc.orc_integral(small, 5, 6, False, True)
    assert isinstance(s, np.ndarray) and np.array_equal(_bits(s), _bits(w[0])) and np.array_equal(_bits(t), _bits(w[2]))
    big = np.full((3000, 3000), 255, np.uint8)                             # 3000 * 3000 * 255 > 2^31: CV_32S sums wrap, like the reference's
    got = cv.integral(dev(big)).cpu().numpy()
    assert got[-1, -1] == np.int64(3000 * 3000 * 255).astype(np.int32) and np.array_equal(got, orc.orc_integral(big, 4)[0])
    with pytest.raises(NotImplementedError):
        cv.integral(dev(rnd((300, 300), np.uint8, 1)), sdepth=5)          # CV_8U -> CV_32F past 2^24 without sqsum / tilted: vector-width dependent on the CPU
    got = cv.integral(dev(rnd((100, 100), np.uint8, 1)), sdepth=5).cpu().numpy()
    assert np.array_equal(got, orc.orc_integral(rnd((100, 100), np.uint8, 1), 5)[0])
