"""Device-resident tracking step (bh_track_refkf): the chain of Tracker._track_wrt_refkf (slam/tracking.py:54-89) and of the matching
half of Tracker._track_local_map (slam/tracking.py:121-136) - cross-check match, distance filter, index extraction, gathers,
CamOnlyBA, inlier count - as one enqueued sequence of kernels, batched over (frame, reference) pairs."""
import ctypes as C
import numpy as np
from . import _lib
from .ba import HUBER_DELTA, CHI2_SIG_VALUE

BH_MEM_HOST, BH_MEM_DEVICE = 0, 1
TRACK_OK, TRACK_FEW_MATCHES, TRACK_BA_FAILED, TRACK_FEW_INLIERS, TRACK_NO_CANDIDATES, TRACK_SKIPPED = 0, 1, 2, 3, 4, 5
COS60 = float(np.cos(np.pi / 3))                   # config.py / slam/tracking.py:19
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


def track_local_map(ctx, q_desc, nq, q_xy, R, t, mp_pts, mp_normals, mp_feat, M, K, see=None, stage1_status=None, width=640, height=480, cos_thr=COS60,
                    d_max=D_HAMMING_MAX, min_matches=15, min_inliers=15, iters=10, huber_delta=HUBER_DELTA, chi2_thr=CHI2_SIG_VALUE, out_stride=None):
    """Tracker._track_local_map (slam/tracking.py:91-162) for B frames from host arrays (bh_track_local_map): q_desc u8[B][Sq][32], nq[B],
    q_xy i64[B][Sq][2]; R f64[B][3][3], t f64[B][3] (frame.R, frame.t), see f64[B][3] or None (= R (0,0,1) normalised); stage1_status int[B]
    or None: the first stage's verdicts - frames that are not TRACK_OK there are skipped (TRACK_SKIPPED), as `if is_ok:` does at :205-212; the candidates in
    the reference's loop order: mp_pts / mp_normals f64[B][Sm][3], mp_feat u8[B][Sm][32], M[B].  Returns a dict: gate_idx, n_gate, qidx,
    tidx (into the gated list), cand (candidate index of each match), dist, n_match, R, t, inlier, n_inlier, status."""
    q_desc = np.ascontiguousarray(q_desc, dtype=np.uint8); mp_feat = np.ascontiguousarray(mp_feat, dtype=np.uint8)
    B, Sq = q_desc.shape[:2]; Sm = mp_feat.shape[1]
    q_xy = np.ascontiguousarray(q_xy, dtype=np.int64).reshape(B, Sq, 2)
    mp_pts = np.ascontiguousarray(mp_pts, dtype=np.float64).reshape(B, Sm, 3); mp_normals = np.ascontiguousarray(mp_normals, dtype=np.float64).reshape(B, Sm, 3)
    nq = np.ascontiguousarray(nq, dtype=np.int32); M = np.ascontiguousarray(M, dtype=np.int32)
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(B, 9); t = np.ascontiguousarray(t, dtype=np.float64).reshape(B, 3)
    see_a = None if see is None else np.ascontiguousarray(see, dtype=np.float64).reshape(B, 3)
    s1 = None if stage1_status is None else np.ascontiguousarray(stage1_status, dtype=np.int32).reshape(B)
    K = np.ascontiguousarray(K, dtype=np.float64)
    S = int(out_stride or max(min(Sq, Sm), 1))
    o = dict(gate_idx=np.zeros((B, Sm), np.int32), n_gate=np.zeros(B, np.int32), qidx=np.zeros((B, S), np.int32), tidx=np.zeros((B, S), np.int32),
             cand=np.zeros((B, S), np.int32), dist=np.zeros((B, S), np.int32), n_match=np.zeros(B, np.int32), R=np.zeros((B, 3, 3)), t=np.zeros((B, 3)),
             inlier=np.zeros((B, S), np.uint8), n_inlier=np.zeros(B, np.int32), status=np.zeros(B, np.int32))
    ctx.check(_lib.lib().bh_track_local_map(ctx.h, _lib.ptr(q_desc), _lib.ptr(nq), _lib.ptr(q_xy), Sq, _lib.ptr(R), _lib.ptr(t), _lib.ptr(see_a), _lib.ptr(s1),
                                            _lib.ptr(mp_pts), _lib.ptr(mp_normals), _lib.ptr(mp_feat), _lib.ptr(M), Sm, B, int(width), int(height), float(cos_thr),
                                            int(d_max), int(min_matches), int(min_inliers), _lib.ptr(K), int(iters), float(huber_delta), float(chi2_thr),
                                            _lib.ptr(o["gate_idx"]), _lib.ptr(o["n_gate"]), _lib.ptr(o["qidx"]), _lib.ptr(o["tidx"]), _lib.ptr(o["cand"]),
                                            _lib.ptr(o["dist"]), _lib.ptr(o["n_match"]), _lib.ptr(o["R"]), _lib.ptr(o["t"]), _lib.ptr(o["inlier"]),
                                            _lib.ptr(o["n_inlier"]), _lib.ptr(o["status"]), S, BH_MEM_HOST))
    o["inlier"] = o["inlier"].astype(bool)
    return o


def track_local_map_device(ctx, q_desc, nq, q_xy, q_stride, R, t, see, mp_pts, mp_normals, mp_feat, M, m_stride, B, K, out, out_stride, stage1_status=None, width=640, height=480,
                           cos_thr=COS60, d_max=D_HAMMING_MAX, min_matches=15, min_inliers=15, iters=10, huber_delta=HUBER_DELTA, chi2_thr=CHI2_SIG_VALUE):
    """Device pointers in and out (ints; `see` may be 0 / None: derived from R on the device; stage1_status: the int[B] verdicts of the first
    stage on the device or 0 / None - frames that are not TRACK_OK there are skipped as `if is_ok:` does, slam/tracking.py:205-212; out: gate_idx, n_gate, qidx, tidx, cand, dist,
    n_match, R, t, inlier, n_inlier, status): enqueued on the context's stream behind whatever produced R and t, no synchronisation."""
    K = np.ascontiguousarray(K, dtype=np.float64)
    v = C.c_void_p
    ctx.check(_lib.lib().bh_track_local_map(ctx.h, v(q_desc), v(nq), v(q_xy), int(q_stride), v(R), v(t), v(see) if see else None, v(stage1_status) if stage1_status else None, v(mp_pts), v(mp_normals), v(mp_feat),
                                            v(M), int(m_stride), int(B), int(width), int(height), float(cos_thr), int(d_max), int(min_matches), int(min_inliers),
                                            _lib.ptr(K), int(iters), float(huber_delta), float(chi2_thr), v(out["gate_idx"]), v(out["n_gate"]), v(out["qidx"]),
                                            v(out["tidx"]), v(out["cand"]), v(out["dist"]), v(out["n_match"]), v(out["R"]), v(out["t"]), v(out["inlier"]),
                                            v(out["n_inlier"]), v(out["status"]), int(out_stride), BH_MEM_DEVICE))
