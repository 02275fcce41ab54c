"""Local-BA window selection and problem construction over the C ABI.

Mirrors CovisibilityGraph.get_local_map(kf, flag_inner_outer=True, flag_with_input_kf=False) — slam/covisibility_graph.py:51-78 —
and the graph walk of optimize_local — :188-218: which keyframes enter the window and which are held fixed, which map points
are seen by more than one of them, and the (keyframe, point, pixel) edge list that feeds BundleAdjustment / bh_ba_set_problem."""
import numpy as np
from . import _lib


def window_problem(ctx, nbr_start, nbr_id, nbr_common, obs_start, obs_mp, obs_px, n_mp, kf, min_common=15):
    nbr_start = np.ascontiguousarray(nbr_start, dtype=np.int32); nbr_id = np.ascontiguousarray(nbr_id, dtype=np.int32)
    nbr_common = np.ascontiguousarray(nbr_common, dtype=np.int32)
    obs_start = np.ascontiguousarray(obs_start, dtype=np.int32); obs_mp = np.ascontiguousarray(obs_mp, dtype=np.int32)
    obs_px = np.ascontiguousarray(obs_px, dtype=np.int64).reshape(-1, 2)
    K = len(nbr_start) - 1
    cap_kf = K + 1; cap_e = max(1, len(obs_mp)); cap_pt = max(1, min(int(n_mp), len(obs_mp)))
    win_kf = np.zeros(cap_kf, np.int32); win_fixed = np.zeros(cap_kf, np.uint8); pt_mp = np.zeros(cap_pt, np.int32)
    e_pose = np.zeros(cap_e, np.int32); e_pt = np.zeros(cap_e, np.int32); e_px = np.zeros((cap_e, 2), np.int64)
    counts = np.zeros(5, np.int32)
    ctx.check(_lib.lib().bh_ba_window_problem(ctx.h, K, _lib.ptr(nbr_start), _lib.ptr(nbr_id), _lib.ptr(nbr_common), _lib.ptr(obs_start),
                                              _lib.ptr(obs_mp), _lib.ptr(obs_px), int(n_mp), int(kf), int(min_common), cap_kf,
                                              _lib.ptr(win_kf), _lib.ptr(win_fixed), cap_pt, _lib.ptr(pt_mp), cap_e, _lib.ptr(e_pose),
                                              _lib.ptr(e_pt), _lib.ptr(e_px), _lib.ptr(counts)))
    nw, n_in, n_out, npt, ne = (int(c) for c in counts)
    return dict(win_kf=win_kf[:nw].copy(), win_fixed=win_fixed[:nw].astype(bool), n_inner=n_in, n_outer=n_out, pt_mp=pt_mp[:npt].copy(),
                edge_pose=e_pose[:ne].copy(), edge_point=e_pt[:ne].copy(), edge_px=e_px[:ne].copy())
