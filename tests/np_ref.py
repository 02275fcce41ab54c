"""Independent numpy / pure-Python second opinions for single stages of the oracle (SURVEY.md §8c "independent
cross-checks that ARE possible here").  Written from the algorithm definitions, not from oracle/*.c, so that a typo in
the C restatement shows up as a disagreement.  Slow: only for small inputs."""
from fractions import Fraction
import numpy as np

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
          (-3, 1), (-2, 2), (-1, 3)]


def gray(bgr):
    b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def _round_half_even(fr):
    fl = fr.numerator // fr.denominator
    rem = fr - fl
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl % 2 == 1):
        return fl + 1
    return fl


def _axis_coeffs(src, dst):
    """Exact-rational version of the INTER_LINEAR_EXACT coefficient rule (double arithmetic is exact enough that the
    two agree unless a product lands within 1e-13 of a rounding boundary)."""
    out = []
    scale = 1.0 / (dst / src)                      # the double the library uses
    for d in range(dst):
        f = Fraction(scale) * (Fraction(d) + Fraction(1, 2)) - Fraction(1, 2)
        i = f.numerator // f.denominator
        if i >= 0 and src > 1:
            if i < src - 1:
                c1 = _round_half_even((f - i) * 256)
                out.append((i, 256 - c1, c1))
            else:
                out.append((src - 1, 256, 0))
        else:
            out.append((0, 256, 0))
    return out


def resize_linear_exact(img, dw, dh):
    sh, sw = img.shape
    cx, cy = _axis_coeffs(sw, dw), _axis_coeffs(sh, dh)
    out = np.zeros((dh, dw), np.uint8)
    im = img.astype(np.int64)
    for y, (r0, a0, a1) in enumerate(cy):
        r1 = min(r0 + 1, sh - 1)
        for x, (c0, b0, b1) in enumerate(cx):
            c1 = min(c0 + 1, sw - 1)
            h0 = b0 * im[r0, c0] + b1 * im[r0, c1]
            h1 = b0 * im[r1, c0] + b1 * im[r1, c1]
            out[y, x] = (a0 * h0 + a1 * h1 + 32768) >> 16
    return out


def fast_is_corner(img, x, y, t):
    v = int(img[y, x])
    ring = [int(img[y + dy, x + dx]) for dx, dy in CIRCLE]
    for sign in (1, -1):
        flags = [(v - p) * sign > t for p in ring]
        for s in range(16):
            if all(flags[(s + k) % 16] for k in range(9)):
                return True
    return False


def fast_score_bruteforce(img, x, y, t):
    """0 if not a corner at threshold t; else the largest threshold at which it still is one."""
    if not fast_is_corner(img, x, y, t):
        return 0
    T = t
    while T < 255 and fast_is_corner(img, x, y, T + 1):
        T += 1
    return T


def hamming_matrix(q, t):
    qb = np.unpackbits(q, axis=1).astype(np.int32); tb = np.unpackbits(t, axis=1).astype(np.int32)
    return (qb[:, None, :] != tb[None, :, :]).sum(-1)


def match_crosscheck(q, t):
    if len(q) == 0 or len(t) == 0:
        return np.zeros((0, 3), np.int64)
    D = hamming_matrix(q, t)                       # [nq][nt]
    tq = D.argmin(axis=0)                          # first minimum per train
    td = D.min(axis=0)
    best = {}
    for j in range(len(t)):
        i = int(tq[j])
        if i not in best or td[j] < best[i][1]:
            best[i] = (j, int(td[j]))
    return np.array([(i, best[i][0], best[i][1]) for i in sorted(best)], np.int64).reshape(-1, 3)


def gauss7_int_kernel():
    x = np.arange(7) - 3.0
    k = np.exp(-0.5 * x * x / 4.0).astype(np.float32)
    k = (k * np.float64(1.0 / k.astype(np.float64).sum())).astype(np.float32)
    return np.rint(k.astype(np.float32) * np.float32(256)).astype(np.int64)


def gauss_blur7(img):
    k = gauss7_int_kernel()
    p = np.pad(img.astype(np.int64), 3, mode="reflect")          # numpy 'reflect' == BORDER_REFLECT_101
    h, w = img.shape
    hor = sum(k[i] * p[:, i:i + w] for i in range(7))
    ver = sum(k[i] * hor[i:i + h, :] for i in range(7))
    return np.clip((ver + 32768) >> 16, 0, 255).astype(np.uint8)


def harris(img, x, y):
    im = img.astype(np.int64)
    a = b = c = 0
    for dy in range(-3, 4):
        for dx in range(-3, 4):
            yy, xx = y + dy, x + dx
            ix = (im[yy, xx + 1] - im[yy, xx - 1]) * 2 + (im[yy - 1, xx + 1] - im[yy - 1, xx - 1]) + (im[yy + 1, xx + 1] - im[yy + 1, xx - 1])
            iy = (im[yy + 1, xx] - im[yy - 1, xx]) * 2 + (im[yy + 1, xx - 1] - im[yy - 1, xx - 1]) + (im[yy + 1, xx + 1] - im[yy - 1, xx + 1])
            a += ix * ix; b += iy * iy; c += ix * iy
    f = np.float32
    scale = f(1.0) / (f(4) * f(7) * f(255.0))
    s4 = scale * scale * scale * scale
    fa, fb, fc = f(a), f(b), f(c)
    return ((fa * fb - fc * fc) - (f(0.04) * (fa + fb)) * (fa + fb)) * s4


def ic_moments(img, x, y):
    im = img.astype(np.int64)
    m10 = m01 = 0
    for v in range(-15, 16):
        for u in range(-15, 16):
            # the 749-pixel disc OpenCV builds: |u| <= umax[|v|]
            if abs(u) <= UMAX[abs(v)]:
                m10 += u * im[y + v, x + u]; m01 += v * im[y + v, x + u]
    return m01, m10


UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def project(R, t, X, K):
    pc = R.T @ (X - t)
    return np.array([K[0] * pc[0] / pc[2] + K[2], K[1] * pc[1] / pc[2] + K[3]])
