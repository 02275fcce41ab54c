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
