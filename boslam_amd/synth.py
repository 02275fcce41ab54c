"""Seeded synthetic RGB-D inputs.  Data generation only — no hot-path arithmetic here.

The reference has no synthetic frame source (frames come from a RealSense, camera.py:131-137, or a pickle,
slam/main.py:53-55); tests and bench.py therefore render a textured plane through the reference's default
pinhole intrinsics (config.py:36-41) with a plain numpy bilinear warp.  BA problems follow the recipe of the
reference's only synthetic generator, tests/ba_test.py:166-210, scaled up as SURVEY.md §8d states.
"""
import numpy as np

# config.py:36-41
FX = FY = 384.239013671875
CX, CY = 322.432373046875, 239.6533203125
BASELINE = 0.06
WIDTH, HEIGHT = 640, 480

_TEX = None


def _texture():
    """2048x2048 texture: 5000 overlapping random rectangles, 30..160 px log-uniform sides (seed 2024).

    Chosen so that every one of the 8 pyramid levels yields more FAST candidates than 2*n_l at nfeatures=1000
    (both retainBest stages are exercised and each frame produces exactly 1000 keypoints, BASELINE config 2)."""
    global _TEX
    if _TEX is None:
        rng = np.random.default_rng(2024)
        t = np.full((2048, 2048), 128, np.uint8)
        for _ in range(5000):
            w = int(np.exp(rng.uniform(np.log(30), np.log(160))))
            h = int(np.exp(rng.uniform(np.log(30), np.log(160))))
            x = int(rng.integers(-w // 2, 2048)); y = int(rng.integers(-h // 2, 2048))
            t[max(y, 0):y + h, max(x, 0):x + w] = rng.integers(0, 256)
        _TEX = t.astype(np.float32)
    return _TEX


def render_frame(k, traj="config2", width=WIDTH, height=HEIGHT):
    """Frame k of the plane-at-2m sequence -> (bgr u8[H][W][3], depth u16[H][W])."""
    tex = _texture()
    if traj == "config1":
        tx, ty = 0.005 * k, 0.0
    else:
        tx, ty = 0.002 * k, 0.001 * k
    ppm = 370.0                                  # texture pixels per metre
    u = np.arange(width, dtype=np.float64)
    v = np.arange(height, dtype=np.float64)
    X = 2.0 * (u - CX) / FX + tx                 # plane z = 2 m, R = I
    Y = 2.0 * (v - CY) / FY + ty
    sx = X * ppm + 640.0
    sy = Y * ppm + 480.0
    x0 = np.floor(sx).astype(np.int64); fx = (sx - x0).astype(np.float32)
    y0 = np.floor(sy).astype(np.int64); fy = (sy - y0).astype(np.float32)
    x0 = np.clip(x0, 0, 2046); y0 = np.clip(y0, 0, 2046)
    t00 = tex[np.ix_(y0, x0)]; t01 = tex[np.ix_(y0, x0 + 1)]
    t10 = tex[np.ix_(y0 + 1, x0)]; t11 = tex[np.ix_(y0 + 1, x0 + 1)]
    top = t00 * (1 - fx)[None, :] + t01 * fx[None, :]
    bot = t10 * (1 - fx)[None, :] + t11 * fx[None, :]
    img = top * (1 - fy)[:, None] + bot * fy[:, None]
    img = img + np.random.default_rng(2025 + 7919 * k).normal(0.0, 4.0, size=img.shape).astype(np.float32)
    gains = np.array([1.0, 0.9, 0.8], dtype=np.float32)            # B, G, R
    bgr = np.clip(np.rint(img[:, :, None] * gains[None, None, :]), 0, 255).astype(np.uint8)
    depth = np.full((height, width), 2000, dtype=np.uint16)
    rng = np.random.default_rng(2026 + 104729 * k)        # ~5 % of the pixels without depth, in 12 coherent holes
    for _ in range(12):
        w = int(rng.integers(20, 60)); h = int(rng.integers(15, 45))
        x = int(rng.integers(0, width - w)); y = int(rng.integers(0, height - h))
        depth[y:y + h, x:x + w] = 0
    return np.ascontiguousarray(bgr), depth


def render_batch(n, traj="config2", start=0):
    bgr = np.empty((n, HEIGHT, WIDTH, 3), dtype=np.uint8)
    depth = np.empty((n, HEIGHT, WIDTH), dtype=np.uint16)
    for i in range(n):
        bgr[i], depth[i] = render_frame(start + i, traj)
    return bgr, depth


def random_descriptors(seed, n):
    """utils.py:53-55 int2orb-style synthetic descriptors: u8[n][32]."""
    return np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)
