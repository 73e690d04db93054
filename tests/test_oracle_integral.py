"""oracle/integral.c (cv::integral restated channel by channel, order of the floating-point additions kept) against cv::integral itself (oracle/_ref): every depth
triple of the reference's table (sumpixels.dispatch.cpp:383-406), with and without the squared and the tilted sum, 1-4 channels, widths 1 / 2 / 3 and ragged ones.
Bit for bit -- float sums included (compared as raw bits, so a -0 counts)."""
import numpy as np
import pytest

import orc

# (source dtype, sum depth, squared-sum depth)
TRIPLES = [(np.uint8, 4, 6), (np.uint8, 4, 5), (np.uint8, 4, 4), (np.uint8, 5, 6), (np.uint8, 5, 5), (np.uint8, 6, 6), (np.uint16, 6, 6), (np.int16, 6, 6),
           (np.float32, 5, 6), (np.float32, 5, 5), (np.float32, 6, 6), (np.float64, 6, 6)]
SIZES = [(1, 1), (1, 7), (7, 1), (5, 2), (2, 5), (3, 3), (9, 33), (40, 67), (64, 64), (31, 130)]


def source(shape, dtype, seed):
    rng = np.random.default_rng(seed)
    if dtype in (np.float32, np.float64):
        a = (rng.standard_normal(shape) * np.exp2(rng.integers(-12, 12, shape))).astype(dtype)       # exponents apart: every addition rounds
        a.flat[:: 7] = 0; a.flat[3:: 11] = -0.0
        return a
    info = np.iinfo(dtype)
    return rng.integers(info.min, int(info.max) + 1, shape).astype(dtype)


def bits(a):
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize]) if a.dtype.kind == "f" else a


def same(a, b):
    return a is None and b is None or np.array_equal(bits(a), bits(b))


needs_ref = pytest.mark.skipif(orc.load_ref() is None, reason="oracle/_ref is not built")


@needs_ref
@pytest.mark.parametrize("dtype,sdepth,sqdepth", TRIPLES)
def test_restatement_is_the_reference(dtype, sdepth, sqdepth):
    n = 0
    for (h, w) in SIZES:
        for cn in (1, 2, 3, 4):
            for (sq, tl) in ((False, False), (True, False), (False, True), (True, True)):
                if sqdepth != 6 and not sq:
                    continue                                                            # the squared sum's depth only matters when there is one
                if tl and cn > 1 and (h, w) not in ((5, 2), (9, 33)):
                    continue
                src = source((h, w, cn) if cn > 1 else (h, w), dtype, 100 * h + w + cn)
                got = orc.orc_integral(src, sdepth, sqdepth, sq, tl)
                want = orc.ref_integral(src, sdepth, sqdepth, sq, tl)
                for g, r, name in zip(got, want, ("sum", "sqsum", "tilted")):
                    assert same(g, r), (name, h, w, cn, sq, tl, None if g is None else np.argwhere(bits(g) != bits(r))[:4])
                n += 1
    assert n > 40


def test_known_answers():
    """a 3 x 3 image by hand: sum and tilted sum as imgproc.hpp:2470-2490 defines them"""
    src = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.uint8)
    S, Q, T = orc.orc_integral(src, 4, 6, True, True)
    assert S.tolist() == [[0, 0, 0, 0], [0, 1, 3, 6], [0, 5, 12, 21], [0, 12, 27, 45]]
    assert Q[-1, -1] == float((src.astype(np.int64) ** 2).sum())
    # tilted(X, Y) = sum of I(x, y) over y < Y, |x - X + 1| <= Y - y - 1
    want = np.zeros((4, 4), np.int64)
    for Y in range(4):
        for X in range(4):
            want[Y, X] = sum(int(src[y, x]) for y in range(3) for x in range(3) if y < Y and abs(x - X + 1) <= Y - y - 1)
    assert T.tolist() == want.tolist()


def test_int_sums_wrap():
    """CV_32S squared sums of a 300 x 300 image of 255s pass 2^31: the reference's int arithmetic wraps, so does the restatement"""
    src = np.full((300, 300), 255, np.uint8)
    S, Q, _ = orc.orc_integral(src, 4, 4, True, False)
    assert S[-1, -1] == 300 * 300 * 255 and int(Q[-1, -1]) == ((300 * 300 * 255 * 255 + 2 ** 31) % 2 ** 32) - 2 ** 31


def test_unknown_triple_is_refused():
    o = orc.oracle()
    a = np.zeros((2, 2), np.float32); s = np.zeros((3, 3), np.int32)
    assert o.orc_integral(5, 4, 6, orc.P(a), orc.step(a), orc.P(s), orc.step(s), None, orc.c_sz(0), None, orc.c_sz(0), 2, 2, 1) == -1
