"""Pins oracle/color.c (CPU only): the reference's adler32 known answers (test_color.cpp:2820-2895,
Imgproc_cvtColor_BE: RNG(0) 263x255 8UC3 -> cvtColor -> adler32) and the real reference."""
import zlib

import numpy as np
import pytest

import refpatterns as rp

# test_color.cpp:2847-2895: code -> adler32 of the 8U result on the RNG(0) image (3-channel source cases)
BE_HASHES = {7: 0x416bd44a,   # COLOR_RGB2GRAY  (:2847)
             6: 0x3008c6b8}   # COLOR_BGR2GRAY  (:2849)




def test_gray_known_answer_hashes(orc, ref):
    src = orc.ref_rng_fill((255, 263, 3), np.uint8, 0, 0, 255)
    for code, want in BE_HASHES.items():
        got = orc.orc_cvtColor(src, code)
        assert zlib.adler32(got.tobytes()) == want, hex(zlib.adler32(got.tobytes()))


def test_golden_input_hash_pinned():
    """the RNG(0) input itself is committed (golden) so the KAT also runs where the real reference is absent"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "color_rng0.npz"))
    import orc as o
    for code, want in BE_HASHES.items():
        assert zlib.adler32(o.orc_cvtColor(g["src"], code).tobytes()) == want


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_oracle_matches_real_reference(orc, ref, dtype):
    hi = {np.uint8: 256, np.uint16: 65536, np.float32: 1.0}[dtype]
    for (w, h) in [(1, 1), (7, 3), (64, 5), (263, 31), (1000, 4)]:
        for code in range(12):
            scn = {0: 3, 1: 4, 2: 3, 3: 4, 4: 3, 5: 4, 6: 3, 7: 3, 8: 1, 9: 1, 10: 4, 11: 4}[code]
            dcn = {0: 4, 1: 3, 2: 4, 3: 3, 4: 3, 5: 4, 6: 1, 7: 1, 8: 3, 9: 4, 10: 1, 11: 1}[code]
            src = orc.ref_rng_fill((h, w, scn) if scn > 1 else (h, w), dtype, 1000 + code + w, 0, hi)
            want = orc.ref_cvtColor(src, code, dcn)
            got = orc.orc_cvtColor(src, code)
            if dtype == np.float32 and dcn == 1:
                # widths below one SIMD register run the reference's scalar tail, whose contraction is the
                # compiler's choice; the lane formula (FMA chain) is what the oracle states
                assert np.max(np.abs(got - want)) <= 1.2e-7 * max(1.0, float(np.max(np.abs(want)))), (w, h, code)
                if w >= 64:
                    assert np.array_equal(got[:, :64], want[:, :64]), (w, h, code)
            else:
                assert np.array_equal(got, want), (w, h, code, dtype)
