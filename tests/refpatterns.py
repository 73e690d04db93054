"""Deterministic inputs restated from the reference's own tests (test infrastructure)."""
import numpy as np


def cv_rng_next(state):
    """cv::RNG::next (core/include/opencv2/core/operations.hpp: MWC, CV_RNG_COEFF 4164903690)."""
    state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
    return state, state & 0xFFFFFFFF


def smooth_bitexact_pattern(rows, cols, cn):
    """The sin / checker / ramp / constant test image of test_smooth_bitexact.cpp:88-109 (CV_8U)."""
    _, r = cv_rng_next(0x123456789abcdef)
    rnd_u8 = r & 0xFF
    j = np.arange(rows, dtype=np.int64)[:, None]
    i = np.arange(cols, dtype=np.int64)[None, :]
    tl = (np.sin((i + 1) * np.pi / 256.) * np.sin((j + 1) * np.pi / 256.) * np.sin((cn + 4) * np.pi / 8.) + 1.) * 128.
    tr = ((i // 128 + j // 128) % 2) * 250 + (j // 128) % 2
    bl = (i // 128) * (85 - j // 256 * 40) * ((j // 128) % 2) + (7 - i // 128) * (85 - j // 256 * 40) * ((j // 128 + 1) % 2)
    br = np.full((rows, cols), float(rnd_u8))
    top = np.where(i < cols // 2, tl, tr.astype(np.float64))
    bot = np.where(i < cols // 2, bl.astype(np.float64), br)
    val = np.where(j < rows // 2, top, bot)
    img = (val.astype(np.int64) & 0xFF).astype(np.uint8)          # (uint8_t)val
    return np.repeat(img[:, :, None], cn, axis=2) if cn > 1 else img


NP_PAD = {0: dict(mode="constant", constant_values=0), 1: dict(mode="edge"), 2: dict(mode="symmetric"),
          3: dict(mode="wrap"), 4: dict(mode="reflect")}


def eval_fixed(src, kx, ky, border, shift_total=16):
    """test_smooth_bitexact.cpp:37-50 eval<uint8_t,8>() applied to every pixel after copyMakeBorder (:115):
    (sum_j ky[j] * sum_i kx[i] * p + 2^15) >> 16, saturated."""
    kx = np.asarray(kx, np.int64)
    ky = np.asarray(ky, np.int64)
    rx, ry = len(kx) // 2, len(ky) // 2
    a = src if src.ndim == 3 else src[:, :, None]
    pad = np.pad(a, ((ry, ry), (rx, rx), (0, 0)), **NP_PAD[border]).astype(np.int64)
    h, w = a.shape[:2]
    acc = np.zeros(a.shape, np.int64)
    for j in range(len(ky)):
        line = np.zeros(a.shape, np.int64)
        for i in range(len(kx)):
            line += pad[j:j + h, i:i + w] * kx[i]
        acc += line * ky[j]
    out = np.clip((acc + (1 << (shift_total - 1))) >> shift_total, 0, 255).astype(np.uint8)
    return out if src.ndim == 3 else out[:, :, 0]
