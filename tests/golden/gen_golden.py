"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libocvref.so, built from
/root/reference by oracle/ref/Makefile).  Run in the build container:

    make -C oracle/ref -j8 && python tests/golden/gen_golden.py

Inputs come from the reference's own cv::RNG(seed).fill(UNIFORM) (ts default seed 809564, ts.cpp:883)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402


def gaussian_u8():
    out = {}
    i = 0
    for cn in (1, 2, 3, 4):
        for (w, h) in [(16, 9), (37, 23), (64, 32), (5, 3), (2, 2)]:
            for ksize, border in [(3, 4), (5, 4), (5, 0), (5, 1), (3, 2), (5, 3), (7, 4), (9, 2)]:
                if i % 3 and (w, h) != (37, 23):
                    i += 1
                    continue
                shape = (h, w, cn) if cn > 1 else (h, w)
                src = orc.ref_rng_fill(shape, np.uint8, 809564 + i, 0, 256)
                dst = orc.ref_GaussianBlur(src, ksize, 0, 0, border | 16)
                k = len([x for x in out if x.startswith("src")])
                out[f"src{k}"], out[f"dst{k}"] = src, dst
                out[f"ksize{k}"], out[f"border{k}"] = ksize, border
                i += 1
    out["n"] = len([x for x in out if x.startswith("src")])
    np.savez_compressed(os.path.join(HERE, "gaussian_u8.npz"), **out)
    print("gaussian_u8.npz:", out["n"], "cases")


def color_rng0():
    """test_color.cpp:2826: RNG(0).fill(263x255 CV_8UC3, UNIFORM, 0, 255) -- the input of Imgproc_cvtColor_BE"""
    src = orc.ref_rng_fill((255, 263, 3), np.uint8, 0, 0, 255)
    np.savez_compressed(os.path.join(HERE, "color_rng0.npz"), src=src,
                        gray_bgr=orc.ref_cvtColor(src, 6, 1), gray_rgb=orc.ref_cvtColor(src, 7, 1))
    print("color_rng0.npz")


if __name__ == "__main__":
    assert orc.load_ref() is not None, "build oracle/_ref first"
    gaussian_u8()
    color_rng0()
