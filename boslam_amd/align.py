"""Rigid alignment of matched 3-D points (bh_align_se3_ransac): the loop-closure verification stage slam/loop_closing.py:17-19 declares
(`_similiarity_transformation(self) -> (R, t)`) and leaves empty.  RANSAC over 3-point sets + Horn / Kabsch closed form, batched."""
import numpy as np
from . import _lib

BH_MEM_HOST = 0
ALIGN_OK, ALIGN_FEW_POINTS, ALIGN_NO_MODEL = 0, 1, 2


def align_se3_ransac_batch(ctx, P, Q, n=None, iterations=200, inlier_thr=0.05, confidence=0.99, min_inliers=15):
    """P, Q f64[B][S][3] (q ~ R p + t), n[B] pairs per problem (default: all S).  Returns dict(R[B][3][3], t[B][3], inlier bool[B][S],
    n_inlier[B], status[B])."""
    P = np.ascontiguousarray(P, dtype=np.float64); Q = np.ascontiguousarray(Q, dtype=np.float64)
    assert P.ndim == 3 and P.shape == Q.shape and P.shape[2] == 3
    B, S = P.shape[:2]
    n = np.full(B, S, np.int32) if n is None else np.ascontiguousarray(n, dtype=np.int32)
    R = np.zeros((B, 3, 3)); t = np.zeros((B, 3)); inl = np.zeros((B, max(S, 1)), np.uint8); ni = np.zeros(B, np.int32); st = np.zeros(B, np.int32)
    if B and S:
        ctx.check(_lib.lib().bh_align_se3_ransac(ctx.h, _lib.ptr(P), _lib.ptr(Q), _lib.ptr(n), B, S, int(iterations), float(inlier_thr), float(confidence),
                                                 int(min_inliers), _lib.ptr(R), _lib.ptr(t), _lib.ptr(inl), _lib.ptr(ni), _lib.ptr(st), BH_MEM_HOST))
    else:
        R[:] = np.eye(3); st[:] = ALIGN_FEW_POINTS
    return dict(R=R, t=t, inlier=inl[:, :S].astype(bool), n_inlier=ni, status=st)
