"""GPU parity: solvePnPRansac / Rodrigues (HIP) vs the CPU oracle — same RANSAC trace, inlier set and BIT-IDENTICAL rvec / tvec
(the refinement sums in point order and both sides use detmath.h) — and the config-1 plumbing harness (counterpart of
experiments/pnp_two_way_tracking.py)."""
import numpy as np
import pytest

import oracle
from boslam_amd import cv2_shim, frontend, pnp, synth

pytestmark = pytest.mark.gpu
K4 = np.array([synth.FX, synth.FY, synth.CX, synth.CY])
KM = np.array([[synth.FX, 0, synth.CX], [0, synth.FY, synth.CY], [0, 0, 1.0]])


def _problem(rng, n, outliers=0.25, noise=0.5):
    R = synth._rodrigues(rng.normal(0, 0.3, 3)); t = rng.normal(0, 0.3, 3) + [0, 0, 0.5]
    X = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1, 1, n), rng.uniform(1.5, 5, n)], 1)
    pc = X @ R.T + t
    uv = np.stack([synth.FX * pc[:, 0] / pc[:, 2] + synth.CX, synth.FY * pc[:, 1] / pc[:, 2] + synth.CY], 1) + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outliers
    uv[bad] += rng.normal(0, 50, (int(bad.sum()), 2))
    return X, uv, R, t


def test_pnp_ransac_batch_matches_oracle(ctx):
    rng = np.random.default_rng(1)
    sizes = [200, 50, 8, 600, 5, 37, 1000, 4]
    probs = [_problem(rng, n) for n in sizes]
    S = max(sizes)
    obj = np.zeros((len(sizes), S, 3)); img = np.zeros((len(sizes), S, 2))
    for b, (X, uv, _, _) in enumerate(probs):
        obj[b, :len(X)] = X; img[b, :len(X)] = uv
    rv, tv, inl, ni, ok = pnp.solve_pnp_ransac_batch(ctx, obj, img, sizes, K4)
    for b, (X, uv, R, t) in enumerate(probs):
        o_ok, orv, otv, oinl = oracle.solve_pnp_ransac(X, uv, K4)
        assert ok[b] == o_ok, b
        if not o_ok:
            assert sizes[b] < 5
            continue
        assert np.array_equal(inl[b, :ni[b]], oinl), (b, ni[b], len(oinl))
        assert np.array_equal(rv[b], orv) and np.array_equal(tv[b], otv), (b, rv[b] - orv, tv[b] - otv)      # bit for bit
        if sizes[b] >= 50:
            assert np.abs(oracle.rodrigues(rv[b]) - R).max() < 0.01 and np.abs(tv[b] - t).max() < 0.02


def test_four_point_problems_take_the_p3p_branch(ctx):
    """n == 4 (OpenCV: model_points = 4, SOLVEPNP_P3P, one hypothesis, all four must be inliers; VERDICT r03 missing 7): a batch of four-point
    problems mixed with larger ones - verdicts, inlier lists, rvec and tvec equal to the checker's bit for bit (p3p_small.h is shared as text),
    the generating pose recovered where the data is exact, an inconsistent fourth point and n == 3 refused."""
    rng = np.random.default_rng(11)
    sizes = [4, 4, 60, 4, 4, 3, 4, 4, 4, 5, 4, 4]
    probs = []
    for b, n in enumerate(sizes):
        X, uv, R, t = _problem(rng, n, outliers=0.0, noise=0.0 if b % 2 == 0 else 0.3)
        if b == 3:
            uv[3] += 45.0                                                  # the fourth point contradicts the pose of the first three
        probs.append((X, uv, R, t))
    S = max(sizes)
    obj = np.zeros((len(sizes), S, 3)); img = np.zeros((len(sizes), S, 2))
    for b, (X, uv, _, _) in enumerate(probs):
        obj[b, :len(X)] = X; img[b, :len(X)] = uv
    rv, tv, inl, ni, ok = pnp.solve_pnp_ransac_batch(ctx, obj, img, sizes, K4)
    n_ok4 = 0
    for b, (X, uv, R, t) in enumerate(probs):
        o_ok, orv, otv, oinl = oracle.solve_pnp_ransac(X, uv, K4)
        assert bool(ok[b]) == o_ok, b
        if not o_ok:
            assert ni[b] == 0
            continue
        assert np.array_equal(inl[b, :ni[b]], oinl) and np.array_equal(rv[b], orv) and np.array_equal(tv[b], otv), (b, rv[b] - orv, tv[b] - otv)
        if sizes[b] == 4:
            n_ok4 += 1
            assert ni[b] == 4
            if b % 2 == 0:                                                 # exact data: the pose itself (float32-rounded input: ~1e-6)
                assert np.abs(oracle.rodrigues(rv[b]) - R).max() < 1e-4 and np.abs(tv[b] - t).max() < 1e-4
    assert not ok[3] and not ok[5] and n_ok4 >= 7
    # the shim: cv2's return shapes for n == 4
    X, uv, R, t = probs[0]
    k, rvec, tvec, inliers = cv2_shim.solvePnPRansac(X, uv, KM, np.zeros((8, 1), np.float32), ctx=ctx)
    assert k and rvec.shape == (3, 1) and np.array_equal(inliers.ravel(), [0, 1, 2, 3]) and np.abs(cv2_shim.Rodrigues(rvec)[0] - R).max() < 1e-4


def test_cv2_shim_pnp_and_rodrigues(ctx):
    rng = np.random.default_rng(3)
    X, uv, R, t = _problem(rng, 120)
    ok, rvec, tvec, inliers = cv2_shim.solvePnPRansac(X, uv, KM, np.zeros((8, 1), np.float32), flags=cv2_shim.SOLVEPNP_ITERATIVE, ctx=ctx)
    assert ok and rvec.shape == (3, 1) and tvec.shape == (3, 1) and inliers.shape[1] == 1 and inliers.dtype == np.int32
    Rm = cv2_shim.Rodrigues(rvec)[0]
    assert Rm.shape == (3, 3) and np.abs(Rm - R).max() < 0.01
    assert np.allclose(Rm, oracle.rodrigues(rvec.ravel()), atol=1e-15)
    back = cv2_shim.Rodrigues(Rm)[0]
    assert back.shape == (3, 1) and np.allclose(back.ravel(), rvec.ravel(), atol=1e-12)
    for v in ([0, 0, 0], [np.pi, 0, 0], [0.1, -2.0, 0.7], [1e-9, 0, 0]):
        assert np.allclose(cv2_shim.Rodrigues(np.array(v, float))[0], oracle.rodrigues(np.array(v, float)), atol=1e-15)


def test_config1_two_way_tracking_plumbing(ctx):
    """Counterpart of experiments/pnp_two_way_tracking.py:37-91 on synthetic frames: ORB -> cross-check match (< 20) ->
    two-way solvePnPRansac -> Rodrigues -> fused (R, t).  Every stage runs on the GPU and is compared with the oracle."""
    nfr = 6
    bgr, depth = synth.render_batch(nfr, traj="config1")
    params = frontend.make_params(nfeatures=600)
    kps, desc, n = frontend.orb_detect_describe(ctx, bgr, depth=depth, params=params)
    xy, cloud = frontend.kp_backproject(ctx, kps, n, depth, synth.FX, synth.FY, synth.CX, synth.CY)
    P = oracle.orb_params(nfeatures=600)
    for f in range(1, nfr):
        ok_, od_, _ = oracle.orb_detect_describe(bgr[f], (depth[f] > 0).astype(np.uint8) * 255, P)
        assert np.array_equal(desc[f, :n[f]], od_)
        qi, ti, d, nm = frontend.match_pairs(ctx, desc[f - 1:f], n[f - 1:f], desc[f:f + 1], n[f:f + 1])      # prev = query (:42)
        oq, ot, odist = oracle.match_crosscheck(desc[f - 1, :n[f - 1]], desc[f, :n[f]])
        assert np.array_equal(qi[0, :nm[0]], oq) and np.array_equal(ti[0, :nm[0]], ot)
        good = d[0, :nm[0]] < 20                                                                               # :43
        ip, ic = qi[0, :nm[0]][good], ti[0, :nm[0]][good]
        assert len(ip) >= 10                                                                                   # :45
        a = cloud[f - 1, ip]; b = xy[f, ic].astype(np.float64); a1 = cloud[f, ic]; b1 = xy[f - 1, ip].astype(np.float64)
        ok_ab, rab, tab, in_ab = cv2_shim.solvePnPRansac(a, b, KM, None, ctx=ctx)
        ok_ba, rba, tba, in_ba = cv2_shim.solvePnPRansac(a1, b1, KM, None, ctx=ctx)
        o_ab = oracle.solve_pnp_ransac(a, b, K4); o_ba = oracle.solve_pnp_ransac(a1, b1, K4)
        assert ok_ab and ok_ba and o_ab[0] and o_ba[0]
        assert np.array_equal(in_ab.ravel(), o_ab[3]) and np.array_equal(in_ba.ravel(), o_ba[3])
        assert np.array_equal(rab.ravel(), o_ab[1]) and np.array_equal(tab.ravel(), o_ab[2])
        assert np.array_equal(rba.ravel(), o_ba[1]) and np.array_equal(tba.ravel(), o_ba[2])
        R_ab = cv2_shim.Rodrigues(rab)[0]; R_ba = cv2_shim.Rodrigues(rba)[0]
        R = (R_ab + np.linalg.inv(R_ba)) / 2; t = (tab - tba) / 2                                              # :71-72
        # the camera moved 5 mm in x between frames (plane at 2 m): the fused estimate must say so
        assert np.abs(R - np.eye(3)).max() < 0.02
        tv = np.asarray(t).ravel()
        assert abs(tv[0] + 0.005) < 0.01 and abs(tv[1]) < 0.01 and abs(tv[2]) < 0.02


def test_pnp_hypothesis_trace_matches_oracle_bit_for_bit(ctx):
    """Every intermediate of every RANSAC hypothesis (sample, control points, M'M, its eigen-decomposition, L, rho, betas, R, t,
    inlier count) equals the oracle's as raw 64-bit patterns: the test hook that located the round-1 parity gap, kept as a test."""
    import ctypes as C
    from boslam_amd import _lib
    rng = np.random.default_rng(7)
    for n, outl in ((250, 0.3), (40, 0.5), (6, 0.0)):
        X, uv, _, _ = _problem(rng, n, outliers=outl)
        obj = np.ascontiguousarray(X[None]); img = np.ascontiguousarray(uv[None])
        rv = np.zeros((1, 3)); tv = np.zeros((1, 3)); inl = np.zeros((1, n), np.int32); ni = np.zeros(1, np.int32); ok = np.zeros(1, np.int32)
        tg = np.zeros((1, 32, 512)); nn = np.array([n], np.int32)
        ctx.check(_lib.lib().bh_debug_pnp_trace(ctx.h, _lib.ptr(obj), _lib.ptr(img), _lib.ptr(nn), 1, n, _lib.ptr(K4), 100, 8.0, 0.99,
                                                _lib.ptr(rv), _lib.ptr(tv), _lib.ptr(inl), _lib.ptr(ni), _lib.ptr(ok), _lib.ptr(tg)))
        to = np.zeros((32, 512))
        oracle.lib().orc_pnp_set_trace(to.ctypes.data_as(C.c_void_p), 32)
        o_ok, orv, otv, oinl = oracle.solve_pnp_ransac(X, uv, K4)
        oracle.lib().orc_pnp_set_trace(None, 0)
        assert to[0].any()
        assert np.array_equal(tg[0].view(np.uint64), to.view(np.uint64)), np.argwhere(tg[0] != to)[:5]
        assert bool(ok[0]) == o_ok and np.array_equal(inl[0, :ni[0]], oinl)
        assert np.array_equal(rv[0], orv) and np.array_equal(tv[0], otv)


def test_pnp_many_iterations_and_hopeless_problems(ctx):
    """iterationsCount beyond the former cap of 256 (the sample stream is drawn lazily, hypothesis by hypothesis): a problem with 85 %
    outliers needs hundreds of hypotheses before the iteration budget shrinks; a problem with no consistent model at all runs every one
    of them.  rvec / tvec / inliers equal the oracle's bit for bit."""
    rng = np.random.default_rng(21)
    for n, outl, iters in ((120, 0.85, 700), (60, 1.0, 400), (200, 0.6, 300)):
        X, uv, _, _ = _problem(rng, n, outliers=outl)
        ok, rv, tv, inl = cv2_shim.solvePnPRansac(X, uv, KM, None, iterationsCount=iters, ctx=ctx)
        o_ok, orv, otv, oinl = oracle.solve_pnp_ransac(X, uv, K4, iterations=iters)
        assert ok == o_ok
        if ok:
            assert np.array_equal(inl.ravel(), oinl) and np.array_equal(rv.ravel(), orv) and np.array_equal(tv.ravel(), otv)
