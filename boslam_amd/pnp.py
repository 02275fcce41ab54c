"""Batched PnP-RANSAC over the C ABI (bh_solve_pnp_ransac)."""
import numpy as np
from . import _lib


def solve_pnp_ransac_batch(ctx, obj, img, n, K4, iterations=100, reproj_error=8.0, confidence=0.99):
    """obj f64[B][S][3], img f64[B][S][2], n int[B] -> rvec[B][3], tvec[B][3], inliers int[B][S], n_inliers[B], ok[B]."""
    obj = np.ascontiguousarray(obj, dtype=np.float64); img = np.ascontiguousarray(img, dtype=np.float64)
    B, S = obj.shape[:2]
    n = np.ascontiguousarray(n, dtype=np.int32); K4 = np.ascontiguousarray(K4, dtype=np.float64)
    rv = np.zeros((B, 3)); tv = np.zeros((B, 3)); inl = np.zeros((B, S), np.int32); ni = np.zeros(B, np.int32); ok = np.zeros(B, np.int32)
    ctx.check(_lib.lib().bh_solve_pnp_ransac(ctx.h, _lib.ptr(obj), _lib.ptr(img), _lib.ptr(n), B, S, _lib.ptr(K4), int(iterations),
                                             float(reproj_error), float(confidence), _lib.ptr(rv), _lib.ptr(tv), _lib.ptr(inl),
                                             _lib.ptr(ni), _lib.ptr(ok), _lib.BH_MEM_HOST))
    return rv, tv, inl, ni, ok.astype(bool)
