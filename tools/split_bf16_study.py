"""Design study for SURVEY section 8 a13 / north_star's bf16 MFMA path of cv::matchTemplate on CV_32F images (CPU only, numpy): how many bf16 x bf16
products does a float correlation need to stay within north_star's 1e-4?  A float x = hi + mid + lo with three bf16 terms (8 significant bits each);
the MFMA accumulates in fp32.  Emulated here: bf16 rounding (round-to-nearest-even on the top 16 bits), exact products, fp32 accumulation in blocks of
32 (the K of one v_mfma_f32_32x32x16_bf16 pair), compared with the double-precision correlation the reference's result is judged against."""
import numpy as np

rng = np.random.default_rng(809564)


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split(x, n):
    parts, rest = [], x.astype(np.float32)
    for _ in range(n):
        p = bf16(rest)
        parts.append(p)
        rest = (rest - p).astype(np.float32)
    return parts


def corr_mfma(img, tpl, terms):
    """sum over the listed (i, j) of img_part[i] * tpl_part[j], fp32 accumulation over K blocks of 32"""
    ip, tp = split(img, 3), split(tpl, 3)
    acc = np.zeros(img.shape[0], np.float32)
    K = img.shape[1]
    for k0 in range(0, K, 32):
        blk = np.zeros(img.shape[0], np.float64)
        for (i, j) in terms:
            blk += (ip[i][:, k0:k0 + 32].astype(np.float64) * tp[j][k0:k0 + 32].astype(np.float64)).sum(axis=1)      # products of bf16 are exact in fp32; the 32-term dot is fp32-accumulated in hardware
        acc = (acc + blk.astype(np.float32)).astype(np.float32)
    return acc


for name, gen in [("uniform [0,1)", lambda s: rng.random(s, dtype=np.float32)), ("uniform [0,255]", lambda s: (rng.random(s, dtype=np.float32) * 255).astype(np.float32)),
                  ("zero-mean normal", lambda s: rng.standard_normal(s).astype(np.float32))]:
    tpl = gen(128 * 128)
    wins = gen((256, 128 * 128))                       # 256 window positions, flattened 128 x 128 windows
    want = wins.astype(np.float64) @ tpl.astype(np.float64)
    scale = np.sqrt((wins.astype(np.float64) ** 2).sum(axis=1) * (tpl.astype(np.float64) ** 2).sum())       # the TM_CCORR_NORMED denominator: errors relative to it bound the normed methods
    for label, terms in [("1 product  (hi*hi)", [(0, 0)]), ("3 products (+ hi*mid, mid*hi)", [(0, 0), (0, 1), (1, 0)]),
                         ("6 products (+ mid*mid, hi*lo, lo*hi)", [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)])]:
        got = corr_mfma(wins, tpl, terms).astype(np.float64)
        rel = np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-30))
        reln = np.max(np.abs(got - want) / scale)
        print(f"{name:18s} {label:38s} max |err| / |corr| = {rel:9.2e}   max |err| / (|I| |T|) = {reln:9.2e}")
    f32 = (wins * tpl).astype(np.float32)
    acc = np.zeros(256, np.float32)
    for k0 in range(0, 128 * 128, 32):
        acc = (acc + f32[:, k0:k0 + 32].sum(axis=1, dtype=np.float32)).astype(np.float32)
    print(f"{name:18s} {'plain fp32 products, fp32 accumulation':38s} max |err| / |corr| = {np.max(np.abs(acc - want) / np.maximum(np.abs(want), 1e-30)):9.2e}")
