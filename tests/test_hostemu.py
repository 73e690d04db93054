"""The per-pixel arithmetic headers the HIP kernels include (opencv_amd/csrc/*_math.h), compiled for the host and checked against the
pinned restatement: verifies the lines the GPU runs where no GPU is present (indexing and launch geometry are what the -m gpu tests add)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import orc as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(ROOT, "tests", "hostemu", "hostemu.cpp")
    out = os.path.join(ROOT, "tests", "hostemu", "libhostemu.so")
    hdr = os.path.join(ROOT, "opencv_amd", "csrc", "hsv_math.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I" + os.path.join(ROOT, "opencv_amd", "csrc"), src, "-o", out])
    return ctypes.CDLL(out)


@pytest.mark.parametrize("code", [54, 55, 70, 71])
def test_hsv2bgr_arithmetic(emu, code):
    rng = np.random.default_rng(code)
    swap, full = o._HSV_INV[code]
    for (w, h) in [(1, 1), (31, 3), (32, 2), (70, 5), (641, 200)]:
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for dcn in (3, 4):
            got = np.empty((h, w, dcn), np.uint8)
            emu.emu_hsv2bgr(o.P(src), o.step(src), o.P(got), o.step(got), w, h, dcn, swap, full)
            assert np.array_equal(got, o.orc_cvtHSVtoBGR(src, code, dcn, 8)), (code, w, h, dcn)
    # every (h, s, v) triple once, in the vector body and in the tail
    hsv = np.stack(np.meshgrid(np.arange(256), np.arange(0, 256, 5), np.arange(0, 256, 3), indexing="ij"), axis=-1).reshape(-1, 3).astype(np.uint8)
    for w in (32, 31):
        n = (len(hsv) // w) * w
        src = np.ascontiguousarray(hsv[:n].reshape(-1, w, 3))
        got = np.empty_like(src)
        emu.emu_hsv2bgr(o.P(src), o.step(src), o.P(got), o.step(got), w, src.shape[0], 3, swap, full)
        assert np.array_equal(got, o.orc_cvtHSVtoBGR(src, code, 3, 8)), (code, w)


@pytest.mark.parametrize("code", [52, 69, 60, 73])
def test_hls_arithmetic_exhaustive(emu, code):
    """the HLS lines of hsv_math.h on all 2^24 8-bit inputs, once in the reference's vector body (rows of 4096) and once in its scalar tail (rows of 7), against the
    restatement that tests/test_oracle_hls.py pins to the reference -- the fused / unfused multiply-adds are what decides a few thousand ties"""
    c = np.arange(1 << 24, dtype=np.uint32)
    allc = np.stack([c & 255, (c >> 8) & 255, (c >> 16) & 255], axis=-1).astype(np.uint8)
    fwd = code in o._HLS_FWD
    swap, full = (o._HLS_FWD if fwd else o._HLS_INV)[code]
    n = (1 << 24) // 7 * 7
    for src in (allc.reshape(4096, 4096, 3), allc[:n].reshape(-1, 7, 3)):
        got = np.empty_like(src)
        h, w = src.shape[:2]
        (emu.emu_bgr2hls if fwd else emu.emu_hls2bgr)(o.P(src), o.step(src), o.P(got), o.step(got), w, h, 3, swap, full)
        assert np.array_equal(got, o.orc_cvtColorHxx(src, code, 3)), (code, w)
    rng = np.random.default_rng(code)
    for (w, h, cn) in [(263, 5, 4), (519, 3, 3), (1, 1, 4)]:
        src = rng.integers(0, 256, (h, w, cn if fwd else 3), dtype=np.uint8)
        got = np.empty((h, w, 3 if fwd else cn), np.uint8)
        (emu.emu_bgr2hls if fwd else emu.emu_hls2bgr)(o.P(src), o.step(src), o.P(got), o.step(got), w, h, cn, swap, full)
        assert np.array_equal(got, o.orc_cvtColorHxx(src, code, cn)), (code, w, h, cn)


# ---- the LDS-tile warp kernel (opencv_amd/csrc/warp8.h): plan, staging, box logic and per-pixel arithmetic run thread by thread on the CPU --------
@pytest.fixture(scope="module")
def emu8():
    src = os.path.join(ROOT, "tests", "hostemu", "warp8_emu.cpp")
    out = os.path.join(ROOT, "tests", "hostemu", "libwarp8emu.so")
    hdr = os.path.join(ROOT, "opencv_amd", "csrc", "warp8.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I" + os.path.join(ROOT, "opencv_amd", "csrc"), src, "-o", out])
    lib = ctypes.CDLL(out)
    lib.emu_warp8.restype = ctypes.c_int
    return lib


def _rot(cx, cy, deg, scale):
    """the INVERSE map (dst -> src) of cv::getRotationMatrix2D(center, deg, scale), as the hook receives it"""
    a = np.deg2rad(deg); al, be = scale * np.cos(a), scale * np.sin(a)
    m = np.array([[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy], [0, 0, 1.0]])
    return np.ascontiguousarray(np.linalg.inv(m)[:2])


def _emu_warp(emu8, src, M, dsize, kind, border=0, bval=(0, 0, 0, 0), fetch=0):
    orc = o.oracle()
    orc.orc_bilinearTabI.restype = ctypes.c_void_p
    tab = ctypes.c_void_p(orc.orc_bilinearTabI())
    M = np.ascontiguousarray(M, np.float64)
    want = (o.orc_warpAffine if kind == 0 else o.orc_warpPerspective)(src, M, dsize, 1, border, bval)
    got = np.full_like(want, 0x5A)
    stats = (ctypes.c_longlong * 7)()
    cn = 1 if src.ndim == 2 else src.shape[2]
    rc = emu8.emu_warp8(o.P(src), o.step(src), src.shape[1], src.shape[0], o.P(got), o.step(got), dsize[0], dsize[1], cn, kind,
                        o.P(M), tab, o.P(want), o.step(want), stats, int(border == 0),
                        sum(int(min(max(round(bval[c]), 0), 255)) << (8 * c) for c in range(cn)), fetch)
    return rc, got, want, list(stats)


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_warp8_affine_tiles_on_the_cpu(emu8, cn):
    rng = np.random.default_rng(cn)
    lean_tiles = [0, 0, 0]
    for (sw, sh, dw, dh, deg, sc) in [(640, 360, 640, 360, 7.0, 0.95), (400, 300, 520, 260, -33.0, 1.1), (256, 200, 131, 77, 90.0, 1.0), (512, 128, 512, 128, 0.0, 1.0),
                                      (300, 300, 300, 300, 45.0, 0.6), (640, 480, 1280, 960, 3.0, 2.0), (256, 200, 132, 80, 90.0, 1.0), (400, 300, 400, 300, 33.0, 1.3), (384, 384, 384, 384, 90.0, 1.0), (384, 384, 384, 384, 80.0, 0.9)]:
        shp = (sh, sw) if cn == 1 else (sh, sw, cn)
        src = rng.integers(0, 256, shp, dtype=np.uint8)
        if (sw * cn) % 4:
            continue
        M = _rot(sw / 2.0, sh / 2.0, deg, sc)
        if (dw, dh) != (sw, sh):
            M = M.copy(); M[:, :2] *= sw / dw                                   # dst pixel -> src pixel of a resized canvas
        # fetch bit 1: per-call term tables; bit 2: the lean path (k_warp8_lean1) serves the all-inside tiles of 1-channel images first
        for border, bval, fetch in [(0, (0, 0, 0, 0), 0), (0, (17.4, 200, 3, 255), 1), (1, (0, 0, 0, 0), 2), (4, (0, 0, 0, 0), 3), (0, (9, 0, 0, 0), 7), (2, (0, 0, 0, 0), 7)]:
            rc, got, want, st = _emu_warp(emu8, src, M, (dw, dh), 0, border, bval, fetch)
            if rc != 0:
                continue                                                       # the plan declined (box too large for LDS): the old kernel serves it
            assert np.array_equal(got, want), (cn, sw, sh, dw, dh, deg, border, fetch, int(np.count_nonzero(got != want)))
            if cn in (1, 3) and fetch == 7:
                lean_tiles[0] += st[4]; lean_tiles[1] += st[5]; lean_tiles[2] += st[6]
                if border == 0:
                    assert st[1] == 0 or st[4] == 0, (sw, sh, deg, st)   # BORDER_CONSTANT: where the lean kernel applies it takes every tile, nothing is left to the sampler
            assert st[0] > (0.9 if border == 0 else 0.3) * dw * dh, (cn, deg, border, st)   # BORDER_CONSTANT: only the source's rim is left to the sampler
    assert cn == 4 or (lean_tiles[0] > 20 and lean_tiles[1] > 10 and lean_tiles[2] > 0), lean_tiles


@pytest.mark.parametrize("cn", [1, 3])
def test_warp8_bicubic_tiles_on_the_cpu(emu8, cn):
    """the bicubic tile path of warp8.h (k_warp8_cubic; opt-in on the GPU until it has run there): the plan with the bicubic margin, the grown box, staging, the 4 x 4 taps from
    the tile through byte funnel shifts, perm + dot2 against the Q15 weight pairs in the kernel's LDS layout -- thread by thread against the restatement pinned to the reference"""
    orc = o.oracle()
    orc.orc_warpTabI.restype = ctypes.c_void_p
    tab = ctypes.c_void_p(orc.orc_warpTabI(0))
    emu8.emu_warp8_cubic.restype = ctypes.c_int
    rng = np.random.default_rng(40 + cn)
    served = 0
    for (sw, sh, dw, dh, deg, sc) in [(640, 360, 640, 360, 7.0, 0.95), (400, 300, 520, 260, -33.0, 1.1), (256, 200, 131, 77, 90.0, 1.0), (512, 128, 512, 128, 0.0, 1.0),
                                      (300, 300, 300, 300, 45.0, 0.6), (640, 480, 1280, 960, 3.0, 2.0), (400, 300, 400, 300, 33.0, 1.3), (384, 384, 384, 384, 90.0, 1.0)]:
        if (sw * cn) % 4:
            continue
        src = rng.integers(0, 256, (sh, sw) if cn == 1 else (sh, sw, cn), dtype=np.uint8)
        M = _rot(sw / 2.0, sh / 2.0, deg, sc)
        if (dw, dh) != (sw, sh):
            M = M.copy(); M[:, :2] *= sw / dw
        M = np.ascontiguousarray(M, np.float64)
        for border, bval in [(0, (0, 0, 0, 0)), (0, (17.4, 200, 3, 255)), (1, (0, 0, 0, 0)), (4, (0, 0, 0, 0)), (2, (0, 0, 0, 0))]:
            want = o.orc_warpAffine(src, M, (dw, dh), 2, border, bval)
            got = np.full_like(want, 0x5A)
            stats = (ctypes.c_longlong * 4)()
            rc = emu8.emu_warp8_cubic(o.P(src), o.step(src), sw, sh, o.P(got), o.step(got), dw, dh, cn, o.P(M), tab, o.P(want), o.step(want), stats, int(border == 0),
                                      sum(int(min(max(round(bval[c]), 0), 255)) << (8 * c) for c in range(cn)))
            if rc != 0:
                continue                                                       # the plan declined (box too large for LDS): the other kernels serve it
            st = list(stats)
            assert np.array_equal(got, want), (cn, sw, sh, dw, dh, deg, border, int(np.count_nonzero(got != want)), st)
            assert st[3] < 1000000, st                                         # the coordinates handed to the sampler are the pixels' own
            assert st[0] > (0.85 if border == 0 else 0.3) * dw * dh, (cn, deg, border, st)
            served += 1
    assert served >= 20, served


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_warp8_perspective_tiles_on_the_cpu(emu8, cn):
    rng = np.random.default_rng(10 + cn)
    for (sw, sh, dw, dh, P3) in [(640, 360, 640, 360, [[1.02, 0.03, -20.0], [0.01, 0.98, 15.0], [1e-5, -2e-5, 1.0]]),
                                 (320, 240, 400, 300, [[0.9, -0.2, 30.0], [0.15, 0.85, -10.0], [2e-4, 1e-4, 1.0]]),
                                 (256, 256, 200, 190, [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [3e-3, 0.0, 1.0]])]:
        shp = (sh, sw) if cn == 1 else (sh, sw, cn)
        src = rng.integers(0, 256, shp, dtype=np.uint8)
        rc, got, want, st = _emu_warp(emu8, src, np.array(P3), (dw, dh), 1)
        if rc != 0:
            continue
        assert np.array_equal(got, want), (cn, sw, sh, int(np.count_nonzero(got != want)))
        assert st[0] > 0.3 * dw * dh, (cn, st)


def test_warp8_separable_weights_equal_the_q15_table():
    """warp8.h evaluates the bilinear weights as (Q + 512) >> 10 with Q = (p00 (32 - ax) + p01 ax)(32 - ay) + (p10 (32 - ax) + p11 ax) ay instead of reading
    the 1024-entry Q15 table of initInterTab2D: identical for every table entry and every byte quadruple -- the entries are 32 x the separable products
    except (0, 0) = (32767, 0, 0, 1), which yields the same 8-bit result"""
    orc = o.oracle()
    orc.orc_bilinearTabI.restype = ctypes.c_void_p
    tab = np.ctypeslib.as_array((ctypes.c_short * 4096).from_address(orc.orc_bilinearTabI())).reshape(32, 32, 4).astype(np.int64)
    rng = np.random.default_rng(0)
    p = rng.integers(0, 256, (20000, 4)).astype(np.int64)
    p[:64] = np.array([[a, b, c, d] for a in (0, 255) for b in (0, 255) for c in (0, 255) for d in (0, 255)] * 4)
    for ay in range(32):
        for ax in range(32):
            w = tab[ay, ax]
            want = np.clip((p @ w + (1 << 14)) >> 15, 0, 255)
            q = (p[:, 0] * (32 - ax) + p[:, 1] * ax) * (32 - ay) + (p[:, 2] * (32 - ax) + p[:, 3] * ax) * ay
            assert np.array_equal((q + 512) >> 10, want), (ay, ax)
    # entry (0, 0) against every pair (p00, p11)
    a, d = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    assert np.array_equal((32767 * a + d + (1 << 14)) >> 15, a)


def test_resize8_lean_tiles_on_the_cpu(emu8):
    """k_resize8_lean's per-tile code (warp8.h rzClassify / leanLoad / leanStore / rzRows with the tables of k_resize8_terms) against the pinned restatement of
    cv::resize INTER_LINEAR on CV_8U: up- and downscales, 1 and 3 channels, sizes that leave partial lanes and partial tiles, INTER_AREA's upscale coefficients"""
    emu8.emu_resize8.restype = ctypes.c_int
    rng = np.random.default_rng(5)
    served = 0
    for cn in (1, 3):
        for (sw, sh, dw, dh, interp) in [(480, 270, 960, 540, 1), (480, 270, 720, 405, 1), (640, 360, 480, 270, 1), (400, 300, 1000, 700, 1), (336, 200, 1336, 804, 1),
                                         (480, 270, 960, 540, 3), (300, 200, 452, 301, 1), (128, 64, 512, 256, 1)]:
            if (sw * cn) % 4 or (dw * cn) % 4:
                continue
            src = rng.integers(0, 256, (sh, sw) if cn == 1 else (sh, sw, cn), dtype=np.uint8)
            want = o.orc_resize(src, (dw, dh), interpolation=interp)
            got = np.full_like(want, 0x5A)
            stats = (ctypes.c_longlong * 3)()
            rc = emu8.emu_resize8(o.P(src), o.step(src), sw, sh, o.P(got), o.step(got), dw, dh, cn, ctypes.c_double(dw / sw), ctypes.c_double(dh / sh),
                                  int(interp == 3), stats)
            if rc != 0:
                continue
            assert stats[1] == 0, (cn, sw, sh, dw, dh, list(stats))
            assert np.array_equal(got, want), (cn, sw, sh, dw, dh, interp, int(np.count_nonzero(got != want)), list(stats))
            served += stats[0]
    assert served > 200


# ---- cv::ORB (opencv_amd/csrc/orb_math.h: per-lane arithmetic of the kernels; orb_host.h: the host control flow of orb.hip) ------------------------------
@pytest.fixture(scope="module")
def emuorb():
    src = os.path.join(ROOT, "tests", "hostemu", "orb_emu.cpp")
    out = os.path.join(ROOT, "tests", "hostemu", "liborbemu.so")
    hdrs = [os.path.join(ROOT, "opencv_amd", "csrc", f) for f in ("orb_math.h", "orb_host.h", "orb_pattern.inc")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I" + os.path.join(ROOT, "opencv_amd", "csrc"), src, "-o", out])
    lib = ctypes.CDLL(out)
    lib.emu_orb_border.restype = ctypes.c_long
    lib.emu_orb_fastAtan2.restype = ctypes.c_float
    lib.emu_orb_fastAtan2.argtypes = [ctypes.c_float, ctypes.c_float]
    return lib


ORB_EMU_CASES = [
    (640, 480, 0, {}),
    (333, 222, 7, dict(edgeThreshold=5, nfeatures=2000, fastThreshold=5)),     # samples reach into the reflected ring
    (400, 300, 6, dict(firstLevel=1)),                                        # level 0 is an upscale, the source image sits on level 1
    (640, 480, 5, dict(WTA_K=4, edgeThreshold=19, patchSize=19)),             # random pattern
    (500, 375, 4, dict(WTA_K=3, scaleFactor=1.5, nlevels=5)),
    (97, 61, 8, dict(nlevels=3, edgeThreshold=8, patchSize=9)),
]


def _orb_layout(emuorb, w, h, p):
    out = np.zeros(4 + 4 * 64, np.int32)
    scales = np.zeros(64, np.float32)
    pitch = emuorb.emu_orb_layout(w, h, p["nlevels"], p["firstLevel"], ctypes.c_double(float(np.float32(p["scaleFactor"]))), p["edgeThreshold"], p["patchSize"], o.P(out), o.P(scales))
    return out, scales, pitch


def _orc_pyramid(img, p, blurred=False):
    orc = o.oracle()
    fn = orc.orc_orbPyramidBlurred if blurred else orc.orc_orbPyramid
    fn.restype = ctypes.c_void_p
    out = np.zeros(4 + 4 * 64, np.int32)
    h, w = img.shape
    ptr = fn(o.P(img), o.step(img), w, h, p["nlevels"], ctypes.c_double(float(np.float32(p["scaleFactor"]))), p["edgeThreshold"], p["firstLevel"], p["patchSize"], o.P(out))
    assert ptr
    buf = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(int(out[3]), int(out[2]))).copy()
    orc.orc_free(ctypes.c_void_p(ptr))
    return buf, out


def _orb_random_cases():
    rng = np.random.default_rng(77)
    out = []
    for t in range(12):
        nl = int(rng.integers(1, 8)); fl = int(rng.integers(0, min(3, nl)))
        out.append((int(rng.integers(150, 420)), int(rng.integers(120, 320)), 200 + t,
                    dict(nfeatures=int(rng.integers(100, 1200)), scaleFactor=float(np.round(rng.uniform(1.1, 2.0), 2)), nlevels=nl, edgeThreshold=int(rng.integers(3, 32)), firstLevel=fl,
                         WTA_K=int(rng.choice([2, 3, 4])), scoreType=int(rng.integers(0, 2)), patchSize=int(rng.integers(5, 32)), fastThreshold=int(rng.integers(5, 30)))))
    return out


@pytest.mark.parametrize("w,h,seed,kw", ORB_EMU_CASES + _orb_random_cases())
def test_orb_kernel_lines_against_the_restatement(emuorb, w, h, seed, kw):
    """layout, border pass, Harris + angle per keypoint and descriptor bytes as the kernels compute them, on the restatement's pyramid"""
    p = dict(o.ORB_DEFAULTS, **kw)
    img = o.orb_scene(w, h, seed)
    pyr, lay = _orc_pyramid(img, p)
    elay, scales, pitch = _orb_layout(emuorb, w, h, p)
    assert np.array_equal(lay, elay) and pitch % 64 == 0 and pitch >= lay[2]
    nl, border, bufW, bufH = (int(v) for v in lay[:4])
    rects = lay[4:4 + 4 * nl].reshape(nl, 4)

    # the border pass: interiors in place (every level but firstLevel), rings and the firstLevel interior written by the kernel's threads
    mine = np.full((bufH, pitch), 0xA5, np.uint8)
    covered = np.zeros((bufH, pitch), bool)
    for l, (x, y, lw, lh) in enumerate(rects):
        if l != p["firstLevel"]:
            mine[y:y + lh, x:x + lw] = pyr[y:y + lh, x:x + lw]
        covered[y - border:y + lh + border, x - border:x + lw + border] = True
    for l in range(nl):
        src = img if l == p["firstLevel"] else None
        ran = emuorb.emu_orb_border(o.P(mine), w, h, nl, p["firstLevel"], ctypes.c_double(float(np.float32(p["scaleFactor"]))), p["edgeThreshold"], p["patchSize"], l,
                                    o.P(src) if src is not None else None, o.step(src) if src is not None else ctypes.c_size_t(0))
        assert ran > 0
    assert np.array_equal(mine[:, :bufW][covered[:, :bufW]], pyr[covered[:, :bufW]])
    assert np.all(mine[~covered] == 0xA5)                                        # nothing outside the extended rectangles is touched

    # Harris response and angle of the final keypoints
    kps, desc = o.orc_ORB(img, **kw)
    assert len(kps) > (20 if seed < 200 else 0)                                  # the seeded random parameter sets (seed >= 200) may leave few keypoints
    bl, _ = _orc_pyramid(img, p, blurred=True)
    blp = np.zeros((bufH, pitch), np.uint8); blp[:, :bufW] = bl
    got = np.zeros(2, np.float32)
    d = np.zeros(32, np.uint8)
    for k, want_d in zip(kps, desc):
        l = int(k["octave"])
        s = np.float32(1.0) / scales[l]
        lx, ly = int(np.rint(np.float32(k["x"]) * s)), int(np.rint(np.float32(k["y"]) * s))
        cx, cy = lx + int(rects[l][0]), ly + int(rects[l][1])
        emuorb.emu_orb_score_angle(o.P(mine), pitch, cx, cy, p["patchSize"] // 2, ctypes.c_float(0.04), o.P(got))
        assert got[1].view(np.int32) == k["angle"].view(np.int32), (k, got)
        if p["scoreType"] == 0:
            assert got[0].view(np.int32) == k["response"].view(np.int32), (k, got)
        emuorb.emu_orb_desc(o.P(blp), pitch, cx, cy, ctypes.c_float(k["angle"]), p["patchSize"], p["WTA_K"], o.P(d))
        assert np.array_equal(d, want_d), k


def test_orb_host_tables_and_culls(emuorb):
    orc = o.oracle()
    orc.orc_fastAtan2.restype = ctypes.c_float
    orc.orc_fastAtan2.argtypes = [ctypes.c_float, ctypes.c_float]
    for patch, wta in [(31, 2), (31, 3), (31, 4), (19, 2), (19, 3), (9, 4), (127, 2), (64, 4)]:
        want = np.zeros(1024, np.int32)
        n = orc.orc_orbPattern(patch, wta, o.P(want))
        got = np.zeros(1024, np.int8)
        assert emuorb.emu_orb_pattern(patch, wta, o.P(got)) == n
        assert np.array_equal(got[:n].astype(np.int32), want[:n]), (patch, wta)
    for half in range(1, 64):
        a, b = np.zeros(half + 2, np.int32), np.zeros(half + 2, np.int32)
        orc.orc_orbUmax(half, o.P(a)); emuorb.emu_orb_umax(half, o.P(b))
        assert np.array_equal(a, b), half
    rng = np.random.default_rng(0)
    for y, x in np.concatenate([rng.integers(-70000, 70000, (4000, 2)), [[0, 0], [0, 5], [5, 0], [0, -5], [-5, 0], [3, 3], [-3, 3], [3, -3], [-3, -3]]]):
        assert np.float32(emuorb.emu_orb_fastAtan2(float(y), float(x))).view(np.int32) == np.float32(orc.orc_fastAtan2(float(y), float(x))).view(np.int32), (y, x)
    # retainBest: the same survivors in the same order (std::nth_element of the C++ library on both sides of the product; the restatement is pinned to it)
    for trial in range(200):
        n = int(rng.integers(1, 700))
        kp = np.zeros(n, o.KP_DTYPE)
        kp["x"] = np.arange(n)
        kp["response"] = rng.integers(0, 12, n) if trial % 2 else rng.random(n)
        for npts in {0, 1, n // 3, n // 2, max(n - 1, 0), n, n + 5}:
            a, b = kp.copy(), kp.copy()
            na = emuorb.emu_orb_retainBest(o.P(a), n, npts)
            nb = orc.orc_retainBest(o.P(b), n, npts)
            assert na == nb and a[:na].tobytes() == b[:nb].tobytes(), (trial, n, npts)
    # runByImageBorder: Rect(b, b, w - 2b, h - 2b).contains(Point(cvRound(x), cvRound(y)))
    kp = np.zeros(2000, o.KP_DTYPE)
    kp["x"] = rng.uniform(-5, 105, 2000).astype(np.float32); kp["y"] = rng.uniform(-5, 85, 2000).astype(np.float32)
    kp["x"][:50] = np.float32(9.5); kp["x"][50:100] = np.float32(10.5); kp["y"][100:150] = np.float32(69.5)
    a = kp.copy()
    na = emuorb.emu_orb_runByImageBorder(o.P(a), len(a), 100, 80, 10)
    rx, ry = np.rint(kp["x"]).astype(int), np.rint(kp["y"]).astype(int)
    keep = (rx >= 10) & (rx < 90) & (ry >= 10) & (ry < 70)
    assert na == keep.sum() and a[:na].tobytes() == kp[keep].tobytes()
    assert emuorb.emu_orb_runByImageBorder(o.P(kp.copy()), len(kp), 20, 80, 10) == 0
    # the first cull on 8-byte (response, pixel index) records gives the survivors of the keypoint form in the same order
    for trial in range(60):
        w, h, b = int(rng.integers(30, 200)), int(rng.integers(30, 120)), int(rng.integers(0, 14))
        n = int(rng.integers(1, 3000))
        idx = np.sort(rng.choice(w * h, size=min(n, w * h), replace=False)).astype(np.int32)            # raster order, like the sorted keys
        n = len(idx)
        kp = np.zeros(n, o.KP_DTYPE)
        kp["x"] = idx % w; kp["y"] = idx // w; kp["class_id"] = idx
        kp["response"] = rng.integers(0, 30, n) if trial % 2 else rng.integers(0, 255, n)
        # the border test runs in the collect kernel (x, y are integers there): the list the host sees is already filtered
        keep = (kp["x"] >= b) & (kp["x"] < w - b) & (kp["y"] >= b) & (kp["y"] < h - b) if b > 0 else np.ones(n, bool)
        for npts in (0, 5, n // 4, n // 2, n, n + 3):
            a, ref = kp[keep].copy(), kp.copy()
            na = emuorb.emu_orb_cullCand(o.P(a), len(a), w, npts) if len(a) else 0
            m = emuorb.emu_orb_runByImageBorder(o.P(ref), n, w, h, b)
            assert m == keep.sum()
            nb = orc.orc_retainBest(o.P(ref), m, npts)
            assert na == nb and np.array_equal(a["class_id"][:na], ref["class_id"][:nb]) and np.array_equal(a["response"][:na], ref["response"][:nb]), (trial, npts)


# ---- the 5 x 5 median on columns sorted once per position (opencv_amd/csrc/median5_math.h, networks of median_net.h) ---------------------------------
@pytest.fixture(scope="module")
def emumed():
    src = os.path.join(ROOT, "tests", "hostemu", "median5_emu.cpp")
    out = os.path.join(ROOT, "tests", "hostemu", "libmedian5emu.so")
    hdrs = [os.path.join(ROOT, "opencv_amd", "csrc", f) for f in ("median5_math.h", "median_net.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I" + os.path.join(ROOT, "opencv_amd", "csrc"), src, "-o", out])
    return ctypes.CDLL(out)


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_median5_sorted_columns_lines(emumed, cn):
    """the kernel's per-lane lines on every 16-byte chunk of whole images: equal to the restatement of cv::medianBlur(5)"""
    rng = np.random.default_rng(cn)
    for (w, h) in [(16, 5), (17, 9), (64, 33), (131, 40), (5, 7), (257, 19)]:
        shape = (h, w) if cn == 1 else (h, w, cn)
        imgs = [rng.integers(0, 256, shape, dtype=np.uint8),
                rng.integers(0, 4, shape, dtype=np.uint8) * 85,                                              # heavy ties
                (rng.integers(0, 2, shape, dtype=np.uint8) * 255),                                           # salt and pepper
                np.broadcast_to(np.arange(w, dtype=np.uint8).reshape((1, w) + (() if cn == 1 else (1,))), shape).copy()]
        for img in imgs:
            got = np.full_like(img, 0x5A)
            assert emumed.emu_median5(o.P(img), o.step(img), o.P(got), o.step(got), w, h, cn) == 0
            assert np.array_equal(got, o.orc_medianBlur(img, 5)), (cn, w, h)


# ---- CV_8U cubic / Lanczos resize on 256 x 16 tiles with staged source bytes (opencv_amd/csrc/resize_tab8.h) ---------------------------------------------
@pytest.fixture(scope="module")
def emurt8():
    src = os.path.join(ROOT, "tests", "hostemu", "resize_tab8_emu.cpp")
    out = os.path.join(ROOT, "tests", "hostemu", "libresizetab8emu.so")
    hdr = os.path.join(ROOT, "opencv_amd", "csrc", "resize_tab8.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-I" + os.path.join(ROOT, "opencv_amd", "csrc"), src, "-o", out])
    return ctypes.CDLL(out)


@pytest.mark.parametrize("interp", [2, 4])
def test_resize_tab8_workgroups_on_the_cpu(emurt8, interp):
    """k_resize_tab8's three phases, every workgroup of the grid thread by thread with guarded LDS buffers of the size the host gives them: equal to the
    restatement of cv::resize(INTER_CUBIC / INTER_LANCZOS4) on CV_8U -- vector body and scalar tail of a row, clamped taps at all four edges, up- and
    mild downscales, 1 / 3 / 4 channels; strong minification is reported as left to the other kernels"""
    rng = np.random.default_rng(interp)
    served = 0
    for (sw, sh, dw, dh) in [(96, 54, 192, 108), (97, 61, 203, 131), (100, 80, 150, 170), (300, 200, 261, 170), (64, 48, 333, 111), (33, 7, 35, 9), (640, 360, 1280, 720),
                             (500, 400, 125, 100)]:
        for cn in (1, 3, 4):
            src = rng.integers(0, 256, (sh, sw, cn) if cn > 1 else (sh, sw), dtype=np.uint8)
            want = o.orc_resize(src, (dw, dh), interpolation=interp)
            got = np.full_like(want, 0x77)
            st = (ctypes.c_long * 5)()
            rc = emurt8.emu_resize_tab8(o.P(src), o.step(src), sw, sh, o.P(got), o.step(got), dw, dh, cn, interp, st)
            assert rc in (0, 1), (rc, sw, sh, dw, dh, cn)
            if rc == 0:
                served += 1
                assert np.array_equal(got, want), (sw, sh, dw, dh, cn)
                assert st[1] <= st[3] and st[2] <= st[4]
            else:
                assert (sw, dw) == (500, 125)                                      # 4 x minification: more staged rows than LDS holds
    assert served >= 21


def test_median_networks_are_what_the_generator_writes(tmp_path):
    """opencv_amd/csrc/median_net.h is generated: tools/gen_median_net.py builds every network, verifies it (exhaustively over the 0/1 inputs that satisfy its
    precondition; the unordered 25-input network on samples) and writes the header -- the committed header equals a fresh run"""
    out = tmp_path / "median_net.h"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_median_net.py"), str(out)], stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(ROOT, "opencv_amd", "csrc", "median_net.h")).read()
