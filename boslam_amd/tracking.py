"""Device-resident tracking step (bh_track_refkf): the chain of Tracker._track_wrt_refkf (slam/tracking.py:54-89) and of the matching
half of Tracker._track_local_map (slam/tracking.py:121-136) - cross-check match, distance filter, index extraction, gathers,
CamOnlyBA, inlier count - as one enqueued sequence of kernels, batched over (frame, reference) pairs."""
import ctypes as C
import numpy as np
from . import _lib
from .ba import HUBER_DELTA, CHI2_SIG_VALUE

BH_MEM_HOST, BH_MEM_DEVICE = 0, 1
TRACK_OK, TRACK_FEW_MATCHES, TRACK_BA_FAILED, TRACK_FEW_INLIERS = 0, 1, 2, 3
D_HAMMING_MAX, MIN_MATCHES_CG = 30, 15            # config.py:13,16


def track_refkf(ctx, q_desc, nq, q_xy, t_desc, nt, t_pts, K, d_max=D_HAMMING_MAX, inclusive=False, min_matches=MIN_MATCHES_CG,
                min_inliers=15, iters=10, huber_delta=HUBER_DELTA, chi2_thr=CHI2_SIG_VALUE, out_stride=None):
    """B pairs from host arrays: q_desc u8[B][Sq][32], nq[B], q_xy i64[B][Sq][2] (frame.kp_arr), t_desc u8[B][St][32], nt[B],
    t_pts f64[B][St][3] (kf_ref.cloud_kp / the gated map points).  inclusive=False is tracking.py:57 (`< d_hamming_max`), True is
    tracking.py:121 (`<=`).  Returns a dict of arrays (qidx, tidx, dist, n_match, R, t, inlier, n_inlier, status)."""
    q_desc = np.ascontiguousarray(q_desc, dtype=np.uint8); t_desc = np.ascontiguousarray(t_desc, dtype=np.uint8)
    B, Sq = q_desc.shape[:2]; St = t_desc.shape[1]
    q_xy = np.ascontiguousarray(q_xy, dtype=np.int64).reshape(B, Sq, 2); t_pts = np.ascontiguousarray(t_pts, dtype=np.float64).reshape(B, St, 3)
    nq = np.ascontiguousarray(nq, dtype=np.int32); nt = np.ascontiguousarray(nt, dtype=np.int32)
    K = np.ascontiguousarray(K, dtype=np.float64)
    S = int(out_stride or max(min(Sq, St), 1))
    o = dict(qidx=np.zeros((B, S), np.int32), tidx=np.zeros((B, S), np.int32), dist=np.zeros((B, S), np.int32), n_match=np.zeros(B, np.int32),
             R=np.zeros((B, 3, 3)), t=np.zeros((B, 3)), inlier=np.zeros((B, S), np.uint8), n_inlier=np.zeros(B, np.int32), status=np.zeros(B, np.int32))
    ctx.check(_lib.lib().bh_track_refkf(ctx.h, _lib.ptr(q_desc), _lib.ptr(nq), _lib.ptr(q_xy), Sq, _lib.ptr(t_desc), _lib.ptr(nt), _lib.ptr(t_pts), St, B,
                                        int(d_max), 1 if inclusive else 0, int(min_matches), int(min_inliers), _lib.ptr(K), int(iters),
                                        float(huber_delta), float(chi2_thr), _lib.ptr(o["qidx"]), _lib.ptr(o["tidx"]), _lib.ptr(o["dist"]),
                                        _lib.ptr(o["n_match"]), _lib.ptr(o["R"]), _lib.ptr(o["t"]), _lib.ptr(o["inlier"]), _lib.ptr(o["n_inlier"]),
                                        _lib.ptr(o["status"]), S, BH_MEM_HOST))
    o["inlier"] = o["inlier"].astype(bool)
    return o


def track_refkf_device(ctx, q_desc, nq, q_xy, q_stride, t_desc, nt, t_pts, t_stride, B, K, out, out_stride, d_max=D_HAMMING_MAX, inclusive=False,
                       min_matches=MIN_MATCHES_CG, min_inliers=15, iters=10, huber_delta=HUBER_DELTA, chi2_thr=CHI2_SIG_VALUE):
    """Device pointers in (ints), device pointers out (dict of ints: qidx, tidx, dist, n_match, R, t, inlier, n_inlier, status): enqueued
    on the context's stream, no synchronisation."""
    K = np.ascontiguousarray(K, dtype=np.float64)
    v = C.c_void_p
    ctx.check(_lib.lib().bh_track_refkf(ctx.h, v(q_desc), v(nq), v(q_xy), int(q_stride), v(t_desc), v(nt), v(t_pts), int(t_stride), int(B),
                                        int(d_max), 1 if inclusive else 0, int(min_matches), int(min_inliers), _lib.ptr(K), int(iters),
                                        float(huber_delta), float(chi2_thr), v(out["qidx"]), v(out["tidx"]), v(out["dist"]), v(out["n_match"]),
                                        v(out["R"]), v(out["t"]), v(out["inlier"]), v(out["n_inlier"]), v(out["status"]), int(out_stride), BH_MEM_DEVICE))
