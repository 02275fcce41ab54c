"""bh_align_se3_ransac (the loop-closure alignment slam/loop_closing.py:17-19 declares and leaves empty) against its CPU checker:
bit-identical R, t, inlier masks, counts and verdicts; recovery of planted transforms; degenerate inputs."""
import numpy as np
import pytest

import oracle
from boslam_amd import align, loop_closing, synth

pytestmark = pytest.mark.gpu


def _problem(rng, n, outliers, noise=0.004):
    R0 = synth._rodrigues(rng.normal(0, 0.6, 3)); t0 = rng.normal(0, 1.0, 3)
    P = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(0.5, 6, n)], 1)
    Q = P @ R0.T + t0 + rng.normal(0, noise, (n, 3))
    bad = rng.random(n) < outliers
    Q[bad] += rng.normal(0, 1.0, (int(bad.sum()), 3))
    return P, Q, R0, t0, bad


def test_align_matches_oracle_bitwise_and_recovers_the_transform(ctx):
    rng = np.random.default_rng(7)
    S = 700
    cases = [(3, 0.0), (4, 0.0), (17, 0.2), (64, 0.3), (65, 0.5), (300, 0.4), (700, 0.6), (129, 0.0), (500, 0.75)]
    P = np.zeros((len(cases), S, 3)); Q = np.zeros_like(P); n = np.zeros(len(cases), np.int32); truth = []
    for b, (nn, out) in enumerate(cases):
        p, q, R0, t0, bad = _problem(rng, nn, out)
        P[b, :nn] = p; Q[b, :nn] = q; n[b] = nn; truth.append((R0, t0, bad))
    for thr, iters, mi in ((0.05, 200, 15), (0.02, 50, 3)):
        got = align.align_se3_ransac_batch(ctx, P, Q, n, iterations=iters, inlier_thr=thr, min_inliers=mi)
        for b, (nn, out) in enumerate(cases):
            st, R, t, m, ni = oracle.align_se3_ransac(P[b, :nn], Q[b, :nn], iters, thr, 0.99, mi)
            assert got["status"][b] == st and got["n_inlier"][b] == ni, (b, nn, got["status"][b], st)
            assert np.array_equal(got["R"][b], R) and np.array_equal(got["t"][b], t), (b, nn)
            assert np.array_equal(got["inlier"][b, :nn], m) and not got["inlier"][b, nn:].any()
            if st == 0 and nn >= 17:
                R0, t0, bad = truth[b]
                assert np.abs(R - R0).max() < 5e-3 and np.abs(t - t0).max() < 2e-2
                assert abs(np.linalg.det(R) - 1) < 1e-12 and np.abs(R @ R.T - np.eye(3)).max() < 1e-12
                assert (m & ~bad).sum() >= 0.9 * (~bad).sum()


def test_align_degenerate_inputs(ctx):
    rng = np.random.default_rng(3)
    line = np.outer(np.linspace(0, 1, 40), [1.0, 2.0, 0.5])                    # collinear points fix no rotation
    same = np.tile([0.3, -0.2, 2.0], (40, 1))
    P = np.stack([line, same, rng.uniform(-1, 1, (40, 3)), rng.uniform(-1, 1, (40, 3))]); Q = P.copy()
    Q[2] = rng.uniform(-1, 1, (40, 3))                                         # unrelated clouds: no consistent alignment
    n = np.array([40, 40, 40, 2], np.int32)
    got = align.align_se3_ransac_batch(ctx, P, Q, n, iterations=60, inlier_thr=0.01, min_inliers=10)
    assert list(got["status"]) == [align.ALIGN_NO_MODEL, align.ALIGN_NO_MODEL, align.ALIGN_NO_MODEL, align.ALIGN_FEW_POINTS]
    for b in range(4):
        st, R, t, m, ni = oracle.align_se3_ransac(P[b, :n[b]], Q[b, :n[b]], 60, 0.01, 0.99, 10)
        assert st == got["status"][b] and np.array_equal(R, got["R"][b]) and np.array_equal(t, got["t"][b]) and ni == got["n_inlier"][b]
    empty = align.align_se3_ransac_batch(ctx, np.zeros((0, 5, 3)), np.zeros((0, 5, 3)))
    assert empty["R"].shape == (0, 3, 3)


def test_similarity_transformation_between_two_rendered_frames(ctx):
    """The whole stage on two frames of the synthetic sequence: ORB + back-projection, descriptor matches, alignment.  The recovered
    motion must agree with the pose-only BA of the tracking step between the same frames (two independent estimators)."""
    from boslam_amd import frontend, tracking
    bgr, depth = synth.render_batch(2, start=3)
    params = frontend.make_params(nfeatures=800)
    kps, desc, nk = frontend.orb_detect_describe(ctx, bgr, depth=depth, params=params)
    xy, cloud = frontend.kp_backproject(ctx, kps, nk, depth, synth.FX, synth.FY, synth.CX, synth.CY)

    class F:
        pass
    a, b = F(), F()
    a.des, a.cloud_kp = desc[0, :nk[0]], cloud[0, :nk[0]]
    b.des, b.cloud_kp = desc[1, :nk[1]], cloud[1, :nk[1]]
    good = lambda f: np.isfinite(f.cloud_kp).all(1) & (np.abs(f.cloud_kp).sum(1) > 0)
    a.des, a.cloud_kp = a.des[good(a)], a.cloud_kp[good(a)]
    b.des, b.cloud_kp = b.des[good(b)], b.cloud_kp[good(b)]
    R, t, ni = loop_closing.similarity_transformation(ctx, b, a, inlier_thr=0.03)
    assert R is not None and ni >= 100
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12
    # a nearly static pair of consecutive frames: small rotation, centimetre-scale translation
    assert np.abs(R - np.eye(3)).max() < 0.05 and np.abs(t).max() < 0.2
    class E:
        des = np.zeros((0, 32), np.uint8); cloud_kp = np.zeros((0, 3))
    assert loop_closing.similarity_transformation(ctx, b, E()) == (None, None, 0)


def test_loop_fusion_pairs_match_the_specification(ctx):
    """bh_loop_fuse_pairs (GlobalMapManager._loop_fusion, slam/loop_closing.py:21-26; reference body `pass`, behaviour specified by this project)
    against its numpy restatement oracle.loop_fuse_pairs: pairs, order and survivor flags exact; the planted duplicates are found, look-alike
    descriptors at the wrong place and true duplicates with too different descriptors are not."""
    import oracle
    from boslam_amd import loop_closing, synth
    rng = np.random.default_rng(12)
    Na, Nb = 1400, 1100
    R = synth._rodrigues(np.array([0.2, -0.1, 0.15])); t = np.array([0.5, -0.2, 0.3])
    feat_a = rng.integers(0, 256, (Na, 32), dtype=np.uint8); feat_b = rng.integers(0, 256, (Nb, 32), dtype=np.uint8)
    pt_b = np.stack([rng.uniform(-3, 3, Nb), rng.uniform(-2, 2, Nb), rng.uniform(1, 6, Nb)], 1)
    pt_a = np.stack([rng.uniform(-3, 3, Na), rng.uniform(-2, 2, Na), rng.uniform(1, 6, Na)], 1)
    dup = rng.permutation(min(Na, Nb))[:500]
    for k, j in enumerate(dup):                                      # A[k] duplicates B[j]
        d = feat_b[j].copy()
        for bit in rng.integers(0, 256, rng.integers(0, 9)):
            d[bit >> 3] ^= np.uint8(1 << (bit & 7))
        feat_a[k] = d
        pt_a[k] = R @ pt_b[j] + t + rng.normal(0, 0.01, 3)
    pt_a[:40] += 0.5                                                  # same descriptor, wrong place: not a duplicate
    feat_a[40:80] = rng.integers(0, 256, (40, 32), dtype=np.uint8)   # right place, unrelated descriptor: not matched
    nobs_a = rng.integers(1, 12, Na).astype(np.int32); nobs_b = rng.integers(1, 12, Nb).astype(np.int32)
    pa, pb, ka = loop_closing.loop_fusion(ctx, feat_a, pt_a, nobs_a, feat_b, pt_b, nobs_b, R, t)
    ra, rb, rk = oracle.loop_fuse_pairs(feat_a, pt_a, nobs_a, feat_b, pt_b, nobs_b, R, t)
    assert np.array_equal(pa, ra) and np.array_equal(pb, rb) and np.array_equal(ka, rk)
    assert 380 <= len(pa) <= 420 and not (pa < 80).any() and (np.diff(pa) > 0).all()
    assert all(dup[a] == b for a, b in zip(pa, pb))
    assert np.array_equal(ka, nobs_a[pa] >= nobs_b[pb]) and 0 < ka.sum() < len(ka)
    e = loop_closing.loop_fusion(ctx, feat_a[:0], pt_a[:0], nobs_a[:0], feat_b, pt_b, nobs_b, R, t)
    assert len(e[0]) == 0


def test_pose_graph_optimisation_closes_a_drifting_loop(ctx):
    """bh_pose_graph_optimize (GlobalMapManager._essential_graph_optimization, slam/loop_closing.py:28-29; reference body `pass`, behaviour specified
    by this project) against oracle.pose_graph_optimize (dense Gauss-Newton in numpy): a ring of 150 keyframes whose odometry drifts, one loop
    edge.  Same number of accepted steps, costs and poses equal to solver tolerance; the loop's drift is distributed (pose errors shrink)."""
    import oracle
    from boslam_amd import loop_closing
    rng = np.random.default_rng(2)
    K = 150
    ang = np.linspace(0, 2 * np.pi, K, endpoint=False)
    Rgt = np.stack([oracle._so3_exp(np.array([0.0, a, 0.0])) @ oracle._so3_exp(rng.normal(0, 0.02, 3)) for a in ang])
    tgt = np.stack([4 * np.sin(ang), 0.1 * rng.normal(size=K), 4 * np.cos(ang)], 1)
    rel = lambda i, j: (Rgt[i].T @ Rgt[j], Rgt[i].T @ (tgt[j] - tgt[i]))
    ei, ej, Zr, Zt, w = [], [], [], [], []
    for i in range(K - 1):
        for j in (i + 1, i + 2):
            if j < K:
                Rz, tz = rel(i, j)
                ei.append(i); ej.append(j); Zr.append(Rz @ oracle._so3_exp(rng.normal(0, 0.004, 3))); Zt.append(tz + rng.normal(0, 0.01, 3)); w.append(1.0)
    R0 = [Rgt[0]]; t0 = [tgt[0]]
    for i in range(K - 1):                                        # dead reckoning along the (i, i+1) edges: the drift the loop has to absorb
        k = ei.index(i) if ej[ei.index(i)] == i + 1 else ei.index(i) + 1
        R0.append(R0[-1] @ Zr[k]); t0.append(t0[-1] + R0[-2] @ Zt[k])
    Rz, tz = rel(K - 1, 0)
    ei.append(K - 1); ej.append(0); Zr.append(Rz); Zt.append(tz); w.append(25.0)      # the loop edge (bh_align_se3_ransac's result), trusted more
    fixed = np.zeros(K, bool); fixed[0] = True
    R0 = np.stack(R0); t0 = np.stack(t0)
    Rg, tg, lg, kept, cg = loop_closing.essential_graph_optimization(ctx, R0, t0, fixed, ei, ej, np.stack(Zr), np.stack(Zt), w, iters=8)
    Ro, to, lo, ko = oracle.pose_graph_optimize(R0, t0, fixed, ei, ej, Zr, Zt, w, iters=8)
    # (at the floor of the cost the last micro-step is kept or not depending on the 15th digit: the dense solve and the CG solve may differ by one)
    assert abs(kept - ko) <= 1 and kept >= 3 and cg > 0
    m = min(len(lg), len(lo))
    assert np.allclose(lg[:m], lo[:m], rtol=1e-7, atol=1e-12), (lg, lo)
    assert abs(lg[-1] - lo[-1]) <= 1e-9 * lo[-1]
    assert np.abs(Rg - Ro).max() < 1e-6 and np.abs(tg - to).max() < 1e-6
    assert np.array_equal(Rg[0], R0[0]) and np.array_equal(tg[0], t0[0])                # the gauge stays
    assert lg[-1] < 0.05 * lg[0] and np.abs(tg - tgt).max() < 0.4 * np.abs(t0 - tgt).max()
    # a graph that is already consistent: nothing to do, nothing changes
    Zc = [rel(i, j) for i, j in zip(ei, ej)]
    R1, t1, l1, k1, _ = loop_closing.essential_graph_optimization(ctx, Rgt, tgt, fixed, ei, ej, np.stack([z[0] for z in Zc]), np.stack([z[1] for z in Zc]), w, iters=3)
    assert l1[0] < 1e-20 and np.abs(R1 - Rgt).max() < 1e-9 and np.abs(t1 - tgt).max() < 1e-9
    with pytest.raises(Exception):
        loop_closing.essential_graph_optimization(ctx, Rgt, tgt, np.zeros(K, bool), ei, ej, np.stack([z[0] for z in Zc]), np.stack([z[1] for z in Zc]), w)


def test_global_map_manager_update_closes_a_loop_with_map_frame_clouds(ctx):
    """GlobalMapManager.update end to end in the reference's data model: cloud_kp and MapPoint.pt3d are MAP-frame coordinates
    (camera.py:32-35, slam/covisibility_graph.py:125-131), so the alignment of the loop's two ends is the drift W of the map and the loop edge
    is T_kf^-1 W T_cand.  A ring of keyframes with drifting odometry: the BoW stage must find the keyframe at the start of the ring, the fused
    pairs must be the same physical points, the loop edge must equal the true relative camera pose, and the pose errors must shrink."""
    import oracle
    from boslam_amd import bow, loop_closing
    rng = np.random.default_rng(31)
    K, M, per_kf = 40, 4000, 300
    ang = np.linspace(0, 2 * np.pi, K, endpoint=False)
    Rgt = np.stack([oracle._so3_exp(np.array([0.0, a, 0.0])) @ oracle._so3_exp(rng.normal(0, 0.02, 3)) for a in ang])
    tgt = np.stack([4 * np.sin(ang), 0.05 * rng.normal(size=K), 4 * np.cos(ang)], 1)
    pa_ = rng.uniform(0, 2 * np.pi, M)
    pts = np.stack([6 * np.sin(pa_), rng.uniform(-1, 1, M), 6 * np.cos(pa_)], 1) + rng.normal(0, 0.3, (M, 3))
    feats = rng.integers(0, 256, (M, 32), dtype=np.uint8)
    rel = lambda i, j: (Rgt[i].T @ Rgt[j], Rgt[i].T @ (tgt[j] - tgt[i]))
    ei, ej, Zr, Zt, w = [], [], [], [], []
    for i in range(K - 1):
        Rz, tz = rel(i, i + 1)
        ei.append(i); ej.append(i + 1); Zr.append(Rz @ oracle._so3_exp(rng.normal(0, 0.006, 3))); Zt.append(tz + rng.normal(0, 0.015, 3)); w.append(1.0)
    R0 = [Rgt[0]]; t0 = [tgt[0]]
    for i in range(K - 1):                                        # dead reckoning: the drift the loop has to absorb
        R0.append(R0[-1] @ Zr[i]); t0.append(t0[-1] + R0[-2] @ Zt[i])
    R0 = np.stack(R0); t0 = np.stack(t0)
    drift = np.abs(t0 - tgt).max()
    assert drift > 0.1

    class KF:
        pass
    kfs = []
    for i in range(K):
        seen = np.argsort(np.linalg.norm(pts - tgt[i] * 1.5, axis=1))[:per_kf]          # the points next to this stretch of the ring
        f = KF(); f.id = 700 + i; f.seen = seen
        d = feats[seen].copy()
        flip = rng.random(d.shape) < 0.01
        d ^= (flip * (1 << rng.integers(0, 8, d.shape))).astype(np.uint8)
        f.des = d
        p_cam = (pts[seen] - tgt[i]) @ Rgt[i] + rng.normal(0, 0.003, (per_kf, 3))          # what the camera measures
        f.cloud_kp = p_cam @ R0[i].T + t0[i]                                                # map frame, through the ESTIMATED pose
        kfs.append(f)
    shared = np.intersect1d(kfs[K - 1].seen, kfs[0].seen)
    assert len(shared) > 150
    db = bow.Dbow(synth.vocabulary(k=10, L=4, seed=6), ctx=ctx)
    for f in kfs:
        db.add(f)
    by_id = {f.id: f for f in kfs}
    ids_all = [f.id for f in kfs]

    def map_points(kf):                                             # map points created around kf: its own estimate of the points it sees
        side = 0 if kf.id < 700 + K // 2 else 100000
        return side + kf.seen, feats[kf.seen], kf.cloud_kp, np.full(per_kf, 3 if side else 5, np.int32)

    fixed = np.zeros(K, bool); fixed[0] = True
    gm = loop_closing.GlobalMapManager(ctx, db, by_id.__getitem__, lambda kf: [kf.id - 1, kf.id - 2, kf.id - 3], map_points,
                                       lambda: (ids_all, R0, t0, fixed, ei, ej, np.stack(Zr), np.stack(Zt), w))
    out = gm.update(kfs[K - 1])
    assert out is not None and out["candidate"] == 700 and out["n_inliers"] >= 100
    # the alignment is the map's drift between the two ends, not a camera-to-camera pose
    Wr, Wt = out["R"], out["t"]
    ia = {p: k for k, p in enumerate(kfs[K - 1].seen)}; ib = {p: k for k, p in enumerate(kfs[0].seen)}
    A = np.stack([kfs[K - 1].cloud_kp[ia[p]] for p in shared]); B = np.stack([kfs[0].cloud_kp[ib[p]] for p in shared])
    assert np.abs(A - (B @ Wr.T + Wt)).max() < 0.03
    # fused pairs: the same physical point on both sides, the better observed side (B, 5 observations) survives
    assert len(out["fused"]) >= 0.8 * len(shared)
    assert all(s + 100000 == l for s, l in out["fused"])
    assert {s for s, _ in out["fused"]} <= set(shared.tolist())
    # the loop edge is the true relative camera pose, to the noise of the alignment
    zr, zt = out["loop_edge"]
    Rz, tz = rel(K - 1, 0)
    assert np.abs(zr - Rz).max() < 0.01 and np.abs(zt - tz).max() < 0.03
    assert np.abs(Wr - Rz).max() > 0.1                               # W itself is nowhere near it (what the edge was before the fix)
    ids, Ro, to = out["poses"]
    assert ids == ids_all and np.array_equal(Ro[0], R0[0])
    assert np.abs(to - tgt).max() < 0.4 * drift and out["cost_log"][-1] < 0.1 * out["cost_log"][0]
