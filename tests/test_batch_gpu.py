"""Frame-batched entry points (frames are the unit that shards): N device-resident frames through ONE call must give, frame for
frame, exactly what the single-image hook gives (those are checked against the oracle in the per-function suites) -- including frames that
are views with padding between them and geometries the rolling kernels do not cover (the per-frame generic kernels inside the batch call)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import opencv_amd
    assert torch.cuda.is_available()
    return opencv_amd


def frames_u8(n, h, w, cn=1, seed=0, padded=False):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    shape = (n, h + (3 if padded else 0), w) + ((cn,) if cn > 1 else ())
    t = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=g)
    return t[:, :h] if padded else t


def same(batch, singles):
    for f, one in enumerate(singles):
        assert batch[f].shape == one.shape and torch.equal(batch[f], one), f


@pytest.mark.parametrize("cn,w,h", [(1, 640, 97), (3, 320, 61), (1, 333, 40)])
def test_filter_family_batches(cv, cn, w, h):
    fr = frames_u8(5, h, w, cn, seed=w, padded=(cn == 3))
    same(cv.SobelBatch(fr, cv.CV_16S, 1, 0, 3), [cv.Sobel(f, cv.CV_16S, 1, 0, 3) for f in fr])
    same(cv.SobelBatch(fr, cv.CV_32F, 0, 1, 3, 0.125), [cv.Sobel(f, cv.CV_32F, 0, 1, 3, 0.125) for f in fr])
    same(cv.SobelBatch(fr, cv.CV_16S, 1, 0, -1, borderType=1), [cv.Sobel(f, cv.CV_16S, 1, 0, -1, borderType=1) for f in fr])
    same(cv.boxFilterBatch(fr, -1, (5, 5)), [cv.boxFilter(f, -1, (5, 5)) for f in fr])
    same(cv.boxFilterBatch(fr, -1, (7, 3), (1, 2), False, 2), [cv.boxFilter(f, -1, (7, 3), (1, 2), False, 2) for f in fr])
    k3 = [0.25, 0.5, 0.25]
    same(cv.sepFilter2DBatch(fr, -1, k3, k3), [cv.sepFilter2D(f, -1, k3, k3) for f in fr])
    kx, ky = [0.1, 0.5, 0.2, 0.05, 0.15], [0.7, -0.1, 0.2]
    same(cv.sepFilter2DBatch(fr, cv.CV_32F, kx, ky, delta=0.5), [cv.sepFilter2D(f, cv.CV_32F, kx, ky, delta=0.5) for f in fr])
    same(cv.thresholdBatch(fr, 100.7, 200, 0), [cv.threshold(f, 100.7, 200, 0)[1] for f in fr])
    same(cv.thresholdBatch(fr, 90, 0, 2), [cv.threshold(f, 90, 0, 2)[1] for f in fr])
    with pytest.raises(NotImplementedError):
        cv.thresholdBatch(fr.to(torch.int16), 1, 2, 0)


@pytest.mark.parametrize("dtype,cn", [(torch.uint8, 1), (torch.uint8, 3), (torch.float32, 1)])
def test_geometry_batches(cv, dtype, cn):
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    shape = (4, 120, 161) + ((cn,) if cn > 1 else ())
    fr = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=g)
    if dtype == torch.float32:
        fr = fr.to(torch.float32) / 255
    for dsize, interp in [((80, 60), 1), ((200, 150), 1), ((97, 33), 0), ((80, 60), 3), ((120, 90), 2)]:
        same(cv.resizeBatch(fr, dsize, interpolation=interp), [cv.resize(f, dsize, interpolation=interp) for f in fr])
    M = cv.getRotationMatrix2D((80.0, 60.0), 9.0, 0.9)
    for flags, border in [(1, 0), (0, 1), (1 | cv.WARP_INVERSE_MAP, 4)]:
        same(cv.warpAffineBatch(fr, M, (161, 120), flags, border, 7.0), [cv.warpAffine(f, M, (161, 120), flags, border, 7.0) for f in fr])
    same(cv.warpAffineBatch(fr, M, (700, 300), 1 | cv.WARP_INVERSE_MAP, 1), [cv.warpAffine(f, M, (700, 300), 1 | cv.WARP_INVERSE_MAP, 1) for f in fr])
    P = np.array([[1.05, 0.04, -6.0], [0.03, 0.95, 5.0], [1e-4, -1e-4, 1.0]])
    same(cv.warpPerspectiveBatch(fr, P, (161, 120), 1 | cv.WARP_INVERSE_MAP, 0, 3.0), [cv.warpPerspective(f, P, (161, 120), 1 | cv.WARP_INVERSE_MAP, 0, 3.0) for f in fr])
    # the tap samplers over a batch (k_warp_taps_lds walks the frames' tiles in one grid, the border strips carry their frame): bicubic and Lanczos, maps that leave the source
    for interp in (2, 4):
        for flags, border in [(interp, 0), (interp | cv.WARP_INVERSE_MAP, 4), (interp, 1)]:
            same(cv.warpAffineBatch(fr, M, (161, 120), flags, border, 7.0), [cv.warpAffine(f, M, (161, 120), flags, border, 7.0) for f in fr])
        same(cv.warpAffineBatch(fr, M, (700, 300), interp | cv.WARP_INVERSE_MAP, 0, 9.0), [cv.warpAffine(f, M, (700, 300), interp | cv.WARP_INVERSE_MAP, 0, 9.0) for f in fr])
        same(cv.warpPerspectiveBatch(fr, P, (161, 120), interp | cv.WARP_INVERSE_MAP, 0, 3.0), [cv.warpPerspective(f, P, (161, 120), interp | cv.WARP_INVERSE_MAP, 0, 3.0) for f in fr])
    if dtype == torch.float32:
        f64 = fr.to(torch.float64)
        for interp in (0, 1, 2, 4):
            same(cv.warpAffineBatch(f64, M, (161, 120), interp, 0, 7.0), [cv.warpAffine(f, M, (161, 120), interp, 0, 7.0) for f in f64])


def test_batch_entries_refuse_host_memory(cv):
    with pytest.raises(ValueError):
        cv.SobelBatch(np.zeros((2, 8, 8), np.uint8), cv.CV_16S, 1, 0)


@pytest.mark.parametrize("scn", [3, 4])
def test_fused_cvtcolor_filter2d(cv, scn):
    """cvtColorFilter2DBatch = filter2D(cvtColor(frame, *2GRAY), -1, K) in one pass over the colour frames: bit-identical to
    the two batched calls -- which are pinned to the oracle elsewhere -- for 3x3 / 5x5 float and integer kernels, both channel orders, three
    borders, delta, widths of one and of several 1 KiB strips, and heights that are not a multiple of the segment length"""
    rng = np.random.default_rng(5 + scn)
    ks = [np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float32), (rng.uniform(-3, 10, (3, 3)) / 31.5).astype(np.float32),
          (rng.uniform(-3, 10, (5, 5)) / 87.5).astype(np.float32), np.ones((5, 5), np.float32) / 25]
    codes = [cv.COLOR_BGR2GRAY, cv.COLOR_RGB2GRAY] if scn == 3 else [cv.COLOR_BGRA2GRAY, cv.COLOR_RGBA2GRAY]
    for (n, h, w) in [(2, 37, 64), (3, 101, 1040), (1, 64, 2064), (2, 5, 16)]:
        fr = torch.from_numpy(rng.integers(0, 256, (n, h, w, scn), dtype=np.uint8)).cuda()
        for code in codes:
            gray = cv.cvtColorBatch(fr, code)
            for k in ks:
                for border, delta in [(4, 0.0), (1, 3.5), (0, 0.0), (2, -2.0)]:
                    want = cv.filter2DBatch(gray, -1, k, delta=delta, borderType=border)
                    got = cv.cvtColorFilter2DBatch(fr, code, k, delta=delta, borderType=border)
                    assert torch.equal(got, want), (n, h, w, code, k.shape, border)
    with pytest.raises(NotImplementedError):                   # a width that is not a multiple of 16 is declined, nothing is computed elsewhere
        cv.cvtColorFilter2DBatch(torch.zeros((1, 20, 40, scn), dtype=torch.uint8, device="cuda"), codes[0], ks[0])


def test_gaussian_batch_from_host_memory_is_pipelined(cv):
    """a batch that lives in host memory (page-locked or pageable) crosses PCIe in chunks through two sets of device buffers; the result
    equals the device-resident batch frame for frame, for chunk counts of one, two and several (odd and even), 1 and 3 channels"""
    g = torch.Generator(); g.manual_seed(3)
    for (n, h, w, cn, pinned) in [(5, 1080, 1920, 1, True), (40, 540, 960, 1, True), (3, 300, 400, 3, False), (33, 270, 480, 3, True), (2, 64, 64, 1, False)]:
        shape = (n, h, w) + ((cn,) if cn > 1 else ())
        host = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
        if pinned:
            host = host.pin_memory()
        want = cv.GaussianBlurBatch(host.cuda(), 5).cpu()
        staged0 = cv._lib.lib.mi355cv_stagedBytes()
        got = cv.GaussianBlurBatch(host, 5)
        assert got.device.type == "cpu" and torch.equal(got, want), (n, h, w, cn, pinned)
        assert cv._lib.lib.mi355cv_stagedBytes() - staged0 == 2 * host.numel()          # every byte crossed PCIe once each way


def test_every_batch_entry_takes_host_resident_frames(cv):
    """every single-output batch entry accepts a batch that lives in host memory (page-locked or pageable) and runs it through
    the two-buffer pipeline (rt.h runHostBatch).  The result equals the device-resident call frame for frame; every byte crosses PCIe once each
    way; chunk counts of one, two and several (the chunk is <= 16 frames / 64 MB)."""
    g = torch.Generator(); g.manual_seed(11)
    M = cv.getRotationMatrix2D((200.0, 150.0), 9.0, 0.9)
    P = np.array([[1.05, 0.04, -6.0], [0.03, 0.95, 5.0], [1e-4, -1e-4, 1.0]])
    k3 = [0.25, 0.5, 0.25]
    sharpen = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float32)
    ops = {
        "sobel":   (1, lambda f: cv.SobelBatch(f, cv.CV_16S, 1, 0, 3)),
        "box":     (3, lambda f: cv.boxFilterBatch(f, -1, (5, 5))),
        "sep":     (1, lambda f: cv.sepFilter2DBatch(f, cv.CV_32F, k3, k3, delta=0.5)),
        "thresh":  (3, lambda f: cv.thresholdBatch(f, 100, 255, 0)),
        "resize":  (3, lambda f: cv.resizeBatch(f, (333, 200), interpolation=1)),
        "affine":  (1, lambda f: cv.warpAffineBatch(f, M, (400, 300), 1, 0, 7.0)),
        "persp":   (3, lambda f: cv.warpPerspectiveBatch(f, P, (384, 216), 1 | cv.WARP_INVERSE_MAP, 1)),
        "gray":    (3, lambda f: cv.cvtColorBatch(f, cv.COLOR_BGR2GRAY)),
        "filter":  (1, lambda f: cv.filter2DBatch(f, -1, sharpen)),
        "fused":   (3, lambda f: cv.cvtColorFilter2DBatch(f, cv.COLOR_BGR2GRAY, sharpen)),
        "pyrdown": (1, lambda f: cv.pyrDownBatch(f)),
        "harris":  (1, lambda f: cv.cornerHarrisBatch(f, 2, 3, 0.04)),
    }
    for name, (cn, op) in ops.items():
        for (n, h, w, pinned) in [(37, 300, 400, True), (3, 300, 400, False), (1, 300, 400, True)]:
            shape = (n, h, w) + ((cn,) if cn > 1 else ())
            host = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
            if pinned:
                host = host.pin_memory()
            want = op(host.cuda()).cpu()
            staged0 = cv._lib.lib.mi355cv_stagedBytes()
            got = op(host)
            assert got.device.type == "cpu" and got.dtype == want.dtype and got.shape == want.shape, name
            assert torch.equal(got, want), (name, n, pinned)
            assert cv._lib.lib.mi355cv_stagedBytes() - staged0 == host.numel() + got.numel() * got.element_size(), name
    # a large chunk count with big frames: 1080p, 40 frames -> several chunks in flight behind one another
    host = torch.randint(0, 256, (40, 1080, 1920), dtype=torch.uint8, generator=g).pin_memory()
    assert torch.equal(cv.SobelBatch(host, cv.CV_16S, 0, 1, 3), cv.SobelBatch(host.cuda(), cv.CV_16S, 0, 1, 3).cpu())
    with pytest.raises((NotImplementedError, ValueError)):          # one end in HBM, the other on the host: not a batch the pipeline takes
        cv.thresholdBatch(host[:2], 100, 255, 0, dst=torch.empty((2, 1080, 1920), dtype=torch.uint8, device="cuda"))


def test_multi_output_batch_entries_take_host_resident_frames(cv):
    """SURVEY section 8 f4, the entries with several outputs per frame: buildPyramidBatch (one image per level: rt.h runHostBatchN) and
    matchTemplateBatch on frames that live in host memory equal the device-resident calls; every byte crosses PCIe once each way (the template
    once per chunk); chunk counts of one, two and several"""
    g = torch.Generator(); g.manual_seed(23)
    for (n, h, w, cn, pinned) in [(37, 301, 403, 1, True), (3, 300, 400, 3, False), (1, 64, 48, 1, True), (19, 540, 960, 1, True)]:
        shape = (n, h, w) + ((cn,) if cn > 1 else ())
        host = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
        if pinned:
            host = host.pin_memory()
        want = cv.buildPyramidBatch(host.cuda(), 3)
        staged0 = cv._lib.lib.mi355cv_stagedBytes()
        got = cv.buildPyramidBatch(host, 3)
        assert len(got) == 4 and got[0] is host
        for a, b in zip(got[1:], want[1:]):
            assert a.device.type == "cpu" and torch.equal(a, b.cpu()), (n, h, w, cn)
        assert cv._lib.lib.mi355cv_stagedBytes() - staged0 == host.numel() + sum(l.numel() for l in got[1:])
        again = cv.buildPyramidBatch(host, 3, dst=got)                                          # levels of a previous call reused
        assert all(a is b for a, b in zip(again, got))
    f32 = torch.rand((5, 200, 300), generator=g)
    for a, b in zip(cv.buildPyramidBatch(f32, 2)[1:], cv.buildPyramidBatch(f32.cuda(), 2)[1:]):
        assert torch.equal(a, b.cpu())
    with pytest.raises((NotImplementedError, ValueError)):                                      # levels in HBM for frames on the host
        cv.buildPyramidBatch(host, 3, dst=want)
    # matchTemplateBatch: CV_8U (MFMA path) and CV_32F, template on the host or in HBM
    for (n, h, w, pinned) in [(21, 240, 320, True), (2, 240, 320, False)]:
        host = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, generator=g)
        if pinned:
            host = host.pin_memory()
        templ = host[0, 50:82, 60:108].contiguous()
        for method in (3, 5, 1):
            want = cv.matchTemplateBatch(host.cuda(), templ.cuda(), method).cpu()
            got = cv.matchTemplateBatch(host, templ, method)
            assert got.device.type == "cpu" and torch.equal(got, want), (n, method)
            assert torch.equal(cv.matchTemplateBatch(host, templ.cuda(), method), want)
    hf = torch.rand((6, 120, 160), generator=g).pin_memory()
    tf = hf[2, 30:50, 40:70].contiguous()
    assert torch.equal(cv.matchTemplateBatch(hf, tf, 5), cv.matchTemplateBatch(hf.cuda(), tf.cuda(), 5).cpu())


def test_integral_batch(cv):
    """mi355cv_integralBatch: every frame's sum / squared sum equals the single-image hook's (which the per-function suite pins to the oracle), for CV_32S
    and CV_64F sums, with and without squared sums, frames that are views with padding, widths around the 256-column tile and heights around 16 rows"""
    for (n, h, w) in [(3, 37, 300), (2, 16, 255), (5, 101, 1040), (1, 64, 257), (4, 1, 1)]:
        parent = frames_u8(n, h + 3, w, seed=h + w)
        fr = parent[:, :h]
        for sdepth in (4, 6):
            got, gotq = cv.integralBatch(fr, sqsum=True, sdepth=sdepth)
            plain = cv.integralBatch(fr, sdepth=sdepth)
            for f in range(n):
                want, wantq = cv.integral(fr[f], sqsum=True, sdepth=sdepth)
                assert torch.equal(got[f], want) and torch.equal(gotq[f], wantq) and torch.equal(plain[f], want), (n, h, w, sdepth, f)
    ref = torch.cumsum(torch.cumsum(fr[0].to(torch.int64), 0), 1)
    assert torch.equal(got[0][1:, 1:].to(torch.int64), ref)
    with pytest.raises(ValueError):
        cv.integralBatch(fr.cpu())
