"""bh_track_refkf (slam/tracking.py:54-89 and :121-136 as one device-resident chain) against the composition of the oracle's pieces:
cross-check match -> distance filter -> gather -> CamOnlyBA -> inlier count.  Exact on every index, count and verdict; 1e-4 on the pose."""
import numpy as np
import pytest

import oracle
from boslam_amd import synth, tracking

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _oracle_pair(p, b, d_max, inclusive, min_matches, min_inliers):
    nq, nt = int(p["nq"][b]), int(p["nt"][b])
    qi, ti, d = oracle.match_crosscheck(p["q_desc"][b, :nq], p["t_desc"][b, :nt])
    keep = d <= d_max if inclusive else d < d_max
    qi, ti, d = qi[keep], ti[keep], d[keep]
    out = dict(qidx=qi, tidx=ti, dist=d, n_match=len(qi))
    if len(qi) == 0 or len(qi) < min_matches:
        return dict(out, status=tracking.TRACK_FEW_MATCHES, n_inlier=0)
    o = oracle.pose_only_ba(p["q_xy"][b][qi].astype(np.float64), p["t_pts"][b][ti], p["K"])
    ninl = int(o["inlier"].sum())
    return dict(out, R=o["R"], t=o["t"], inlier=o["inlier"], n_inlier=ninl,
                status=tracking.TRACK_BA_FAILED if o["rc"] != 0 else (tracking.TRACK_FEW_INLIERS if ninl < min_inliers else tracking.TRACK_OK))


def _check(p, got, d_max, inclusive, min_matches=15, min_inliers=15):
    B = len(p["nq"])
    for b in range(B):
        ref = _oracle_pair(p, b, d_max, inclusive, min_matches, min_inliers)
        n = ref["n_match"]
        assert got["n_match"][b] == n, b
        assert np.array_equal(got["qidx"][b, :n], ref["qidx"]) and np.array_equal(got["tidx"][b, :n], ref["tidx"]) and np.array_equal(got["dist"][b, :n], ref["dist"])
        assert got["status"][b] == ref["status"], (b, got["status"][b], ref["status"])
        assert got["n_inlier"][b] == ref["n_inlier"]
        if ref["status"] != tracking.TRACK_FEW_MATCHES:
            assert np.array_equal(got["inlier"][b, :n], ref["inlier"])
            assert np.abs(got["R"][b] - ref["R"]).max() < RTOL * max(1.0, np.abs(ref["R"]).max()) and np.abs(got["t"][b] - ref["t"]).max() < RTOL * max(1.0, np.abs(ref["t"]).max())
    return True


def test_track_refkf_matches_oracle_composition(ctx):
    p = synth.tracking_pairs(B=9, n_ref=600, n_frame=600, seed=5)
    for d_max, inclusive in ((30, False), (30, True), (12, False)):
        got = tracking.track_refkf(ctx, p["q_desc"], p["nq"], p["q_xy"], p["t_desc"], p["nt"], p["t_pts"], p["K"], d_max=d_max, inclusive=inclusive)
        _check(p, got, d_max, inclusive)
    assert (got["status"] == tracking.TRACK_OK).sum() >= 5
    # the tracked pose is the generator's pose (synth.project takes (R, t) camera-in-world, the SBACam convention CamOnlyBA returns)
    got = tracking.track_refkf(ctx, p["q_desc"], p["nq"], p["q_xy"], p["t_desc"], p["nt"], p["t_pts"], p["K"])
    for b in range(9):
        if got["status"][b] == tracking.TRACK_OK:
            assert np.abs(got["R"][b] - p["R"][b]).max() < 1e-2 and np.abs(got["t"][b] - p["t"][b]).max() < 3e-2


def test_track_refkf_verdicts_and_edge_cases(ctx):
    """Too few matches (CamOnlyBA not run), too few inliers, the local-map shape (a few thousand map points against 600 features,
    `<=` filter, tracking.py:121), the boundary distance itself, empty sides."""
    p = synth.tracking_pairs(B=6, n_ref=2500, n_frame=600, seed=11, frac_seen=0.5)
    p["nq"][1] = 0; p["nt"][2] = 0                                            # empty query / empty train
    p["q_desc"][3] = synth.random_descriptors(99, 600)                        # nothing re-observed: only chance matches far above 30 bits
    bad = p["q_xy"][4].copy(); bad[:] = np.random.default_rng(3).integers(0, 480, bad.shape); p["q_xy"][4] = bad      # matches fine, geometry garbage
    got = tracking.track_refkf(ctx, p["q_desc"], p["nq"], p["q_xy"], p["t_desc"], p["nt"], p["t_pts"], p["K"], inclusive=True)
    _check(p, got, 30, True)
    assert got["status"][1] == got["status"][2] == got["status"][3] == tracking.TRACK_FEW_MATCHES and got["n_match"][1] == 0
    assert got["status"][4] in (tracking.TRACK_FEW_INLIERS, tracking.TRACK_BA_FAILED) and got["status"][0] == tracking.TRACK_OK
    # `<` against `<=` at the boundary: plant matches at exactly d_max bits
    q = p["t_desc"][5, :40].copy()
    for k in range(40):
        for bit in range(30):
            q[k, bit >> 3] ^= np.uint8(1 << (bit & 7))
    p["q_desc"][5, :40] = q
    a = tracking.track_refkf(ctx, p["q_desc"], p["nq"], p["q_xy"], p["t_desc"], p["nt"], p["t_pts"], p["K"], inclusive=False)
    c = tracking.track_refkf(ctx, p["q_desc"], p["nq"], p["q_xy"], p["t_desc"], p["nt"], p["t_pts"], p["K"], inclusive=True)
    _check(p, a, 30, False); _check(p, c, 30, True)
    assert c["n_match"][5] > a["n_match"][5] and (c["dist"][5, :c["n_match"][5]] == 30).any() and not (a["dist"][5, :a["n_match"][5]] == 30).any()


def test_live_tracker_equals_the_host_staged_calls(ctx):
    """frontend.LiveTracker (one frame per call, previous frame's descriptors / cloud kept on the device, one verdict back) against the
    three host-mode entry points called one after the other on the same rendered frames: same verdict, same pose, bit for bit."""
    torch = pytest.importorskip("torch")
    from boslam_amd import frontend
    bgr, depth = synth.render_batch(5)
    params = frontend.make_params(nfeatures=600)
    K = np.array([synth.FX, synth.FY, synth.CX, synth.CY, 0.06])
    lt = frontend.LiveTracker(ctx, bgr.shape[1], bgr.shape[2], params, K)
    prev = None
    try:
        for i in range(len(bgr)):
            got = lt.step(bgr[i], depth[i])                      # (the class synchronises on every step itself: its pinned staging pair is reused)
            assert (got is None) == (i == 0)
            if got is not None:
                kps, desc, nk = frontend.orb_detect_describe(ctx, bgr[i][None], depth=depth[i][None], params=params)
                xy, cloud = frontend.kp_backproject(ctx, kps, nk, depth[i][None], K[0], K[1], K[2], K[3])
                ref = tracking.track_refkf(ctx, desc, nk, xy, prev[0], prev[1], prev[2], K)
                assert got["status"] == int(ref["status"][0]) and got["n_match"] == int(ref["n_match"][0]) and got["n_inlier"] == int(ref["n_inlier"][0])
                assert np.array_equal(got["R"], ref["R"][0]) and np.array_equal(got["t"], ref["t"][0])
                prev = (desc, nk, cloud)
            else:
                kps, desc, nk = frontend.orb_detect_describe(ctx, bgr[i][None], depth=depth[i][None], params=params)
                xy, cloud = frontend.kp_backproject(ctx, kps, nk, depth[i][None], K[0], K[1], K[2], K[3])
                prev = (desc, nk, cloud)
    finally:
        lt.close()


def test_live_tracker_graph_replay_equals_the_enqueued_steps(ctx):
    """frontend.LiveTracker(use_graph=True): from the third frame on a step replays a captured HIP graph (staging copies, the launches of ORB,
    back-projection and the tracking step, the verdict copy; one graph per slot parity) instead of enqueuing them.  Same frames through both forms:
    every verdict and pose bit for bit - and the frames differ from step to step, so a graph that had captured data instead of work would show."""
    torch = pytest.importorskip("torch")
    from boslam_amd import frontend
    bgr, depth = synth.render_batch(9)
    params = frontend.make_params(nfeatures=600)
    K = np.array([synth.FX, synth.FY, synth.CX, synth.CY, 0.06])
    order = [0, 1, 2, 3, 2, 5, 6, 7, 8, 4]                       # not monotone: a replay must read the frame that was staged for it
    res = {}
    for form in (False, True):
        lt = frontend.LiveTracker(ctx, bgr.shape[1], bgr.shape[2], params, K, use_graph=form)
        try:
            res[form] = [lt.step(bgr[i], depth[i]) for i in order]
            assert (lt.graphs[0] is not None and lt.graphs[1] is not None) == form
        finally:
            lt.close()
    assert res[False][0] is None and res[True][0] is None
    n_ok = 0
    for a, b in zip(res[False][1:], res[True][1:]):
        assert a["status"] == b["status"] and a["n_match"] == b["n_match"] and a["n_inlier"] == b["n_inlier"]
        assert np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"])
        n_ok += a["status"] == 0
    assert n_ok >= 6


def _oracle_local_map(p, b, d_max=30, min_matches=15, min_inliers=15, see=None):
    """slam/tracking.py:91-136 as the composition of the parts' oracles: gate -> stacked feats / pts3d -> match `<=` -> CamOnlyBA -> verdict."""
    M = int(p["M"][b]); R, t = p["R"][b], p["t"][b]
    sv = see if see is not None else (R @ np.array([0.0, 0.0, 1.0])) / np.linalg.norm(R @ np.array([0.0, 0.0, 1.0]))
    g = oracle.visibility_gate(p["mp_pts"][b, :M], p["mp_normals"][b, :M], R, t, synth.FX, synth.CX, synth.CY, 640, 480, sv)
    out = dict(gate=g)
    if len(g) == 0:
        return dict(out, status=tracking.TRACK_NO_CANDIDATES, n_match=0, n_inlier=0)
    nq = int(p["nq"][b])
    qi, ti, d = oracle.match_crosscheck(p["q_desc"][b, :nq], p["mp_feat"][b, g])
    keep = d <= d_max
    qi, ti, d = qi[keep], ti[keep], d[keep]
    out.update(qidx=qi, tidx=ti, dist=d, n_match=len(qi), cand=g[ti])
    if len(qi) < min_matches:
        return dict(out, status=tracking.TRACK_FEW_MATCHES, n_inlier=0)
    o = oracle.pose_only_ba(p["q_xy"][b][qi].astype(np.float64), p["mp_pts"][b, g][ti], p["K"])
    ninl = int(o["inlier"].sum())
    return dict(out, R=o["R"], t=o["t"], inlier=o["inlier"], n_inlier=ninl,
                status=tracking.TRACK_BA_FAILED if o["rc"] != 0 else (tracking.TRACK_FEW_INLIERS if ninl < min_inliers else tracking.TRACK_OK))


def _check_local_map(p, got, see=None):
    for b in range(len(p["nq"])):
        ref = _oracle_local_map(p, b, see=None if see is None else see[b])
        assert got["n_gate"][b] == len(ref["gate"]) and np.array_equal(got["gate_idx"][b, :len(ref["gate"])], ref["gate"]), b
        assert got["status"][b] == ref["status"], (b, got["status"][b], ref["status"])
        n = ref["n_match"]
        assert got["n_match"][b] == n and got["n_inlier"][b] == ref["n_inlier"], b
        if n:
            assert np.array_equal(got["qidx"][b, :n], ref["qidx"]) and np.array_equal(got["tidx"][b, :n], ref["tidx"]) and np.array_equal(got["dist"][b, :n], ref["dist"])
            assert np.array_equal(got["cand"][b, :n], ref["cand"])
        if ref["status"] in (tracking.TRACK_OK, tracking.TRACK_FEW_INLIERS, tracking.TRACK_BA_FAILED):
            assert np.array_equal(got["inlier"][b, :n], ref["inlier"])
            assert np.abs(got["R"][b] - ref["R"]).max() < 1e-9 and np.abs(got["t"][b] - ref["t"]).max() < 1e-9


def test_track_local_map_matches_oracle_composition(ctx):
    """bh_track_local_map (slam/tracking.py:91-136): indices, masks and verdicts exact, pose to 1e-9, every early return forced."""
    p = synth.local_map_frames(B=7, n_frame=600, n_mp=900, seed=9)
    p["M"][1] = 0                                                                 # no candidates at all
    p["mp_normals"][2] = np.array([0.0, 0.0, 1.0])                                # every normal along the viewing direction: the cos60 test leaves nothing
    p["q_desc"][3] = synth.random_descriptors(123, 600)                           # nothing re-observed: chance matches only -> too few
    bad = p["q_xy"][4].copy(); bad[:] = np.random.default_rng(3).integers(0, 480, bad.shape); p["q_xy"][4] = bad      # matches fine, geometry garbage
    got = tracking.track_local_map(ctx, p["q_desc"], p["nq"], p["q_xy"], p["R"], p["t"], p["mp_pts"], p["mp_normals"], p["mp_feat"], p["M"], p["K"])
    _check_local_map(p, got)
    st = got["status"]
    assert st[1] == st[2] == tracking.TRACK_NO_CANDIDATES and st[3] == tracking.TRACK_FEW_MATCHES
    assert st[4] in (tracking.TRACK_FEW_INLIERS, tracking.TRACK_BA_FAILED) and st[0] == st[5] == st[6] == tracking.TRACK_OK
    # duplicates are real: some matched candidate's map point also sits elsewhere in the candidate list, and the ids come out through `cand`
    b = 0
    n = got["n_match"][b]
    ids = p["mp_id"][b][got["cand"][b, :n]]
    assert len(np.unique(p["mp_id"][b, :p["M"][b]])) < p["M"][b] and (ids >= 0).all()
    # an explicit see_vector (a frame whose pose was set more than once, camera.py:26-29) instead of the one derived from R
    see = np.tile(np.array([0.05, -0.02, 1.0]) / np.linalg.norm([0.05, -0.02, 1.0]), (7, 1))
    got2 = tracking.track_local_map(ctx, p["q_desc"], p["nq"], p["q_xy"], p["R"], p["t"], p["mp_pts"], p["mp_normals"], p["mp_feat"], p["M"], p["K"], see=see)
    _check_local_map(p, got2, see=see)
    # Tracker.update calls _track_local_map only `if is_ok` (slam/tracking.py:205-212): with the first stage's verdicts passed in, the frames it
    # lost are not run at all (also when their pose is still all zero, FEW_MATCHES leaves R / t untouched) and say so; the others are unchanged
    s1 = np.array([0, 0, 0, 0, 0, tracking.TRACK_FEW_MATCHES, tracking.TRACK_FEW_INLIERS], np.int32)
    R0 = p["R"].copy(); R0[5] = 0.0
    for see_k in (None, see):
        got3 = tracking.track_local_map(ctx, p["q_desc"], p["nq"], p["q_xy"], R0, p["t"], p["mp_pts"], p["mp_normals"], p["mp_feat"], p["M"], p["K"],
                                        see=see_k, stage1_status=s1)
        ref3 = got if see_k is None else got2
        assert np.array_equal(got3["status"][:5], ref3["status"][:5]) and (got3["status"][5:] == tracking.TRACK_SKIPPED).all()
        assert (got3["n_gate"][5:] == 0).all() and (got3["n_match"][5:] == 0).all() and (got3["n_inlier"][5:] == 0).all()
        for k in ("gate_idx", "qidx", "tidx", "cand", "dist", "inlier", "n_gate", "n_match", "n_inlier"):
            assert np.array_equal(got3[k][:5], ref3[k][:5]), k
        assert np.array_equal(got3["R"][:5], ref3["R"][:5]) and np.array_equal(got3["t"][:5], ref3["t"][:5])
