"""The one live call of the reference's loop-closure stage.

slam/loop_closing.py is a stub: every stage of GlobalMapManager is `pass` except `_loop_candidate_detection`, which issues
`self.dbow.query(kf.des, [], 1, -1)` (:13-15) and drops the result; `update` would raise on the `None` it unpacks (:31-35) and the
class is never constructed (slam/main.py).  What can be matched is therefore that query.  A loop candidate must not come from
the keyframe's own neighbourhood, so the query here is restricted - through the fork's target_inds argument (slam/bow_db.py:37-39)
- to the database entries outside the keyframe's local window; with an empty exclusion it is exactly the reference's call."""
from . import bow


def loop_candidate_detection(dbow, kf, exclude_frame_ids=(), min_score=0.0):
    """dbow: bow.Dbow (mirror of slam/bow_db.py).  kf: object with .id and .des.  exclude_frame_ids: frame ids of the local window
    (CovisibilityGraph.get_local_map(kf), slam/covisibility_graph.py:51-78) plus kf itself.
    Returns (frame id, score) of the best remaining entry, or (None, None) - the (Id, Score) slam/bow_db.py:41 reads."""
    skip = {dbow.frame2entry[f] for f in exclude_frame_ids if f in dbow.frame2entry}
    if kf.id in dbow.frame2entry:
        skip.add(dbow.frame2entry[kf.id])
    subset = [e for e in sorted(dbow.entry2frame) if e not in skip] if skip else []
    if skip and not subset:
        return None, None
    fid, score = dbow.query(kf, subset_inds=subset)
    if fid is None or score < min_score:
        return None, None
    return fid, score


def similarity_transformation(ctx, kf, cand, d_hamming_max=30, iterations=200, inlier_thr=0.05, min_matches=15, min_inliers=15):
    """GlobalMapManager._similiarity_transformation (slam/loop_closing.py:17-19: `-> (R, t)`, body `pass`): "R, t that transforms start of
    the loop with the end".  kf, cand: objects with .des (u8[N][32]) and .cloud_kp (f64[N][3], the back-projected keypoints of
    camera.py:90-118) - the current keyframe and the loop candidate loop_candidate_detection returned.  Their descriptors are matched as
    the tracker matches them (cross-check, `distance < d_hamming_max`, slam/tracking.py:56-57), the matched 3-D points are aligned by
    bh_align_se3_ransac: cloud_kf ~ R cloud_cand + t (with the reference's map-frame cloud_kp this is the drift of the map between the two
    keyframes, see GlobalMapManager; with camera-frame clouds it is the relative camera pose T_kf^-1 T_cand).  Returns (R, t, n_inliers) or (None, None, 0) - too few matches / no consistent
    alignment (a false BoW candidate)."""
    import numpy as np
    from . import frontend, align
    qd = np.ascontiguousarray(kf.des, dtype=np.uint8).reshape(-1, 32); td = np.ascontiguousarray(cand.des, dtype=np.uint8).reshape(-1, 32)
    if len(qd) == 0 or len(td) == 0:
        return None, None, 0
    qi, ti, d, nm = frontend.match_pairs(ctx, qd[None], [len(qd)], td[None], [len(td)])
    k = int(nm[0])
    keep = d[0, :k] < d_hamming_max
    qi, ti = qi[0, :k][keep], ti[0, :k][keep]
    if len(qi) < min_matches:
        return None, None, 0
    P = np.asarray(cand.cloud_kp, dtype=np.float64).reshape(-1, 3)[ti]; Q = np.asarray(kf.cloud_kp, dtype=np.float64).reshape(-1, 3)[qi]
    o = align.align_se3_ransac_batch(ctx, P[None], Q[None], iterations=iterations, inlier_thr=inlier_thr, min_inliers=min_inliers)
    if o["status"][0] != align.ALIGN_OK:
        return None, None, int(o["n_inlier"][0])
    return o["R"][0], o["t"][0], int(o["n_inlier"][0])


def loop_fusion(ctx, feat_a, pt_a, nobs_a, feat_b, pt_b, nobs_b, R, t, d_hamming_max=30, radius=0.05):
    """GlobalMapManager._loop_fusion(R, t) (slam/loop_closing.py:21-26: "finds duplicated mappoints and fuses them together", body `pass`; the
    behaviour is this project's specification, include/boslam_hip.h): side A = map points around the current keyframe, side B = around the loop
    candidate, (R, t) from similarity_transformation.  Returns (pair_a, pair_b, keep_a): indices into the two sides and which point survives;
    the caller re-points the graph's edges (edges_kf2mps / edges_mp2kfs) from the loser to the survivor."""
    import ctypes as C
    import numpy as np
    from . import _lib
    feat_a = np.ascontiguousarray(feat_a, dtype=np.uint8).reshape(-1, 32); feat_b = np.ascontiguousarray(feat_b, dtype=np.uint8).reshape(-1, 32)
    Na, Nb = len(feat_a), len(feat_b)
    if Na == 0 or Nb == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, bool)
    pt_a = np.ascontiguousarray(pt_a, dtype=np.float64).reshape(Na, 3); pt_b = np.ascontiguousarray(pt_b, dtype=np.float64).reshape(Nb, 3)
    nobs_a = np.ascontiguousarray(nobs_a, dtype=np.int32).reshape(Na); nobs_b = np.ascontiguousarray(nobs_b, dtype=np.int32).reshape(Nb)
    Rm = np.ascontiguousarray(R, dtype=np.float64).reshape(9); tv = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
    cap = min(Na, Nb)
    pa = np.zeros(cap, np.int32); pb = np.zeros(cap, np.int32); ka = np.zeros(cap, np.uint8); n = C.c_int(0)
    ctx.check(_lib.lib().bh_loop_fuse_pairs(ctx.h, _lib.ptr(feat_a), _lib.ptr(pt_a), _lib.ptr(nobs_a), Na, _lib.ptr(feat_b), _lib.ptr(pt_b), _lib.ptr(nobs_b), Nb,
                                            _lib.ptr(Rm), _lib.ptr(tv), int(d_hamming_max), float(radius), _lib.ptr(pa), _lib.ptr(pb), _lib.ptr(ka), C.byref(n), cap, 0))
    k = n.value
    return pa[:k], pb[:k], ka[:k].astype(bool)


def essential_graph_optimization(ctx, R, t, fixed, ei, ej, Zr, Zt, w=None, iters=10):
    """GlobalMapManager._essential_graph_optimization (slam/loop_closing.py:28-29, body `pass`; behaviour specified by this project,
    include/boslam_hip.h: bh_pose_graph_optimize).  R f64[K][3][3], t f64[K][3] keyframe poses (camera in world), fixed bool[K]; edges (ei, ej)
    with measured relative poses Zr f64[E][3][3], Zt f64[E][3] ~ T_i^-1 T_j and weights w.  Returns (R, t, cost_log, steps_kept, cg_iterations)."""
    import ctypes as C
    import numpy as np
    from . import _lib
    R = np.array(R, dtype=np.float64).reshape(-1, 9); K = len(R)
    t = np.array(t, dtype=np.float64).reshape(K, 3)
    fx = np.ascontiguousarray(fixed, dtype=np.uint8).reshape(K)
    ei = np.ascontiguousarray(ei, dtype=np.int32); ej = np.ascontiguousarray(ej, dtype=np.int32); E = len(ei)
    Zr = np.ascontiguousarray(Zr, dtype=np.float64).reshape(E, 9); Zt = np.ascontiguousarray(Zt, dtype=np.float64).reshape(E, 3)
    w = np.ones(E) if w is None else np.ascontiguousarray(w, dtype=np.float64).reshape(E)
    log = np.zeros(iters + 1); kept = C.c_int(0); cg = C.c_int(0)
    ctx.check(_lib.lib().bh_pose_graph_optimize(ctx.h, K, _lib.ptr(R), _lib.ptr(t), _lib.ptr(fx), E, _lib.ptr(ei), _lib.ptr(ej), _lib.ptr(Zr), _lib.ptr(Zt),
                                                _lib.ptr(w), int(iters), _lib.ptr(log), C.byref(kept), C.byref(cg)))
    return R.reshape(K, 3, 3), t, log[:kept.value + 1], kept.value, cg.value


class GlobalMapManager:
    """Mirror of slam/loop_closing.py:8-39 with every stage filled in over the C ABI (the reference has a body only for the first one):
        update(kf) = _loop_candidate_detection -> _similiarity_transformation -> _loop_fusion -> _essential_graph_optimization.
    The map itself stays with the caller's graph; the manager is handed what the stages need:
      dbow        bow.Dbow (mirror of slam/bow_db.py)
      get_kf      frame id -> object with .des, .cloud_kp (the loop candidate's keyframe)
      local_ids   kf -> frame ids of kf's local window (CovisibilityGraph.get_local_map), excluded from the candidates
      map_points  kf -> (ids, feat u8[N][32], pt3d f64[N][3], n_obs int[N]) of the map points seen around kf (both ends of the loop)
      pose_graph  () -> (frame ids, R f64[K][3][3], t f64[K][3], fixed bool[K], ei, ej, Zr, Zt, w): the essential graph as it stands
    update returns None when no loop is accepted, else dict(candidate, R, t, n_inliers, fused = [(survivor id, loser id)], poses = (ids, R, t)).

    ONE convention for (R, t), the reference's data model: `cloud_kp` of a keyframe and `pt3d` of a map point are MAP-frame coordinates
    (camera.py:32-35 `transform2global` rewrites cloud_kp with the frame's pose; slam/covisibility_graph.py:125-131 builds MapPoint.pt3d from
    kf.cloud_kp), and a pose of pose_graph() is camera-in-map, p_map = R p_cam + t (the T of `Z ~ T_i^-1 T_j`).  The alignment of the two ends'
    clouds is then the DRIFT of the map between them, cloud_kf ~ W cloud_cand with W = (R, t) near the identity: _loop_fusion compares map points
    in that frame (|pt_a - (R pt_b + t)|), and the loop edge between the two CAMERAS is Z = T_kf^-1 W T_cand (a point the candidate's camera sees
    at p lies at T_cand p in the candidate's part of the map, at W T_cand p in the current keyframe's part, at T_kf^-1 W T_cand p in its camera)."""

    def __init__(self, ctx, dbow, get_kf, local_ids, map_points, pose_graph, min_score=0.05, loop_weight=25.0):
        self.ctx, self.dbow, self.get_kf, self.local_ids, self.map_points, self.pose_graph = ctx, dbow, get_kf, local_ids, map_points, pose_graph
        self.min_score, self.loop_weight = min_score, loop_weight

    @staticmethod
    def loop_edge(R_kf, t_kf, R_cand, t_cand, R, t):
        """Z = T_kf^-1 W T_cand as (Zr, Zt): the relative camera pose the drift W = (R, t) of the map between the loop's ends implies."""
        import numpy as np
        R_kf = np.asarray(R_kf, dtype=np.float64).reshape(3, 3); R_cand = np.asarray(R_cand, dtype=np.float64).reshape(3, 3)
        t_kf = np.asarray(t_kf, dtype=np.float64).reshape(3); t_cand = np.asarray(t_cand, dtype=np.float64).reshape(3)
        R = np.asarray(R, dtype=np.float64).reshape(3, 3); t = np.asarray(t, dtype=np.float64).reshape(3)
        return R_kf.T @ R @ R_cand, R_kf.T @ (R @ t_cand + t - t_kf)

    def update(self, kf):
        import numpy as np
        fid, score = loop_candidate_detection(self.dbow, kf, self.local_ids(kf), self.min_score)
        if fid is None:
            return None
        cand = self.get_kf(fid)
        R, t, n_inl = similarity_transformation(self.ctx, kf, cand)
        if R is None:
            return None
        ids_a, feat_a, pt_a, nobs_a = self.map_points(kf)
        ids_b, feat_b, pt_b, nobs_b = self.map_points(cand)
        pa, pb, ka = loop_fusion(self.ctx, feat_a, pt_a, nobs_a, feat_b, pt_b, nobs_b, R, t)
        fused = [((ids_a[a], ids_b[b]) if k else (ids_b[b], ids_a[a])) for a, b, k in zip(pa, pb, ka)]
        ids, Rk, tk, fixed, ei, ej, Zr, Zt, w = self.pose_graph()
        ids = list(ids)
        Rk = np.asarray(Rk, dtype=np.float64).reshape(-1, 3, 3); tk = np.asarray(tk, dtype=np.float64).reshape(-1, 3)
        i_kf, i_cand = ids.index(kf.id), ids.index(fid)
        # the loop edge current keyframe -> candidate: the clouds are map-frame, so (R, t) is the map's drift W and Z = T_kf^-1 W T_cand
        zr, zt = self.loop_edge(Rk[i_kf], tk[i_kf], Rk[i_cand], tk[i_cand], R, t)
        ei = list(ei) + [i_kf]; ej = list(ej) + [i_cand]
        Zr = np.concatenate([np.asarray(Zr, dtype=np.float64).reshape(-1, 3, 3), zr[None]], 0)
        Zt = np.concatenate([np.asarray(Zt, dtype=np.float64).reshape(-1, 3), zt[None]], 0)
        w = list(w) + [self.loop_weight]
        Ro, to, log, kept, cg = essential_graph_optimization(self.ctx, Rk, tk, fixed, ei, ej, Zr, Zt, w)
        return dict(candidate=fid, score=score, R=R, t=t, n_inliers=n_inl, fused=fused, poses=(ids, Ro, to), cost_log=log, loop_edge=(zr, zt))
