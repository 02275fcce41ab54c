"""GPU parity: pose-only BA and local BA (HIP, f64) vs the CPU oracle (g2o restatement).
Tolerance: 1e-4 relative on optimised poses / landmarks (BASELINE.json north_star); in practice ~1e-9."""
import numpy as np
import pytest

import oracle
from boslam_amd import ba, optimizer, synth

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def test_pose_only_ba_batch_matches_oracle(ctx):
    cases = [dict(n=200, seed=3, outlier_frac=0.2), dict(n=600, seed=4, outlier_frac=0.05), dict(n=15, seed=5, outlier_frac=0.0),
             dict(n=40, seed=6, outlier_frac=0.4), dict(n=1000, seed=7, outlier_frac=0.1, int_pixels=False)]
    probs = [synth.pose_problem(**c) for c in cases]
    S = max(len(p["pix"]) for p in probs)
    pix = np.zeros((len(probs), S, 2)); pts = np.ones((len(probs), S, 3)); n = np.zeros(len(probs), np.int32)
    for b, p in enumerate(probs):
        n[b] = len(p["pix"]); pix[b, :n[b]] = p["pix"]; pts[b, :n[b]] = p["pts"]
    R, t, chi2, inl, st = ba.pose_only_ba_batch(ctx, pix, pts, n, probs[0]["K"])
    for b, p in enumerate(probs):
        o = oracle.pose_only_ba(p["pix"], p["pts"], p["K"])
        assert st[b] == 0 and o["rc"] == 0
        assert _rel(R[b], o["R"]) < RTOL and _rel(t[b], o["t"]) < RTOL, (b, _rel(R[b], o["R"]), _rel(t[b], o["t"]))
        assert np.allclose(chi2[b, :n[b]], o["chi2"], rtol=1e-6, atol=1e-9)
        assert np.array_equal(inl[b, :n[b]], o["inlier"])
        # and the optimiser actually recovers the pose on the inlier-dominated cases
        if cases[b]["outlier_frac"] <= 0.2 and n[b] >= 100:
            assert np.abs(t[b] - p["t"]).max() < 0.02


def test_camonlyba_shim_signature_and_failure_path(ctx):
    class Cam:                                           # config.py used as the camera object (slam/main.py:23-24)
        fx, fy, cx, cy, baseline = synth.FX, synth.FY, synth.CX, synth.CY, synth.BASELINE
    p = synth.pose_problem(n=120, seed=11)
    ok, R, t, inliers = optimizer.CamOnlyBA(p["pix"].astype(np.int64), p["pts"], Cam, ctx=ctx)
    o = oracle.pose_only_ba(p["pix"], p["pts"], p["K"])
    assert ok and _rel(R, o["R"]) < RTOL and _rel(t, o["t"]) < RTOL
    assert np.array_equal(inliers, np.nonzero(o["inlier"])[0])

    class Cam2:                                          # the other flavour of optimizer.py:27-32
        focal_length, principal_point, baseline = synth.FX, (synth.CX, synth.CY), synth.BASELINE
    ok2, R2, t2, _ = optimizer.CamOnlyBA(p["pix"], p["pts"], Cam2, ctx=ctx)
    assert ok2 and np.array_equal(R2, R)
    bad = optimizer.CamOnlyBA(p["pix"], p["pts"], object(), ctx=ctx)      # no intrinsics -> swallowed exception
    assert bad == (False, None, None, None)


@pytest.mark.parametrize("cfg", [dict(n_kf=6, n_pts=150, seed=1), dict(n_kf=12, n_pts=600, seed=2, outlier_frac=0.05),
                                 dict(n_kf=20, n_pts=2000, seed=3, pix_noise=0.5)])
@pytest.mark.parametrize("verbose", [True, False])
def test_local_ba_matches_oracle(ctx, cfg, verbose):
    pr = synth.ba_problem(**cfg)
    delta = ba.HUBER_DELTA
    o = oracle.ba_optimize(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"], pr["K"], 7, delta,
                           verbose_errors=verbose)
    s = ba.LocalBA(ctx, pr["K"])
    s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    done = s.optimize(7, delta, verbose_errors=verbose)
    R, t = s.poses(); pts = s.points(); chi2 = s.chi2(); lg = s.log()[:done]
    assert done == o["iters_done"]
    assert np.array_equal(lg[:, 2:], o["log"][:, 2:]), "trial counts / accept flags differ"
    assert np.allclose(lg[:, :2], o["log"][:, :2], rtol=1e-6)
    assert _rel(R, o["R"]) < RTOL and _rel(t, o["t"]) < RTOL and _rel(pts, o["pts"]) < RTOL
    assert np.allclose(chi2, o["chi2"], rtol=1e-5, atol=1e-8)
    assert np.array_equal(chi2 > 7.815, o["chi2"] > 7.815)
    # fixed keyframes are returned unchanged (covisibility_graph.py:222-225 writes them back too)
    nf = int(pr["fixed"].sum())
    assert np.allclose(R[:nf], pr["R"][:nf], atol=1e-14) and np.array_equal(t[:nf], pr["t"][:nf])
    s.close()


def test_local_ba_is_bitwise_reproducible_and_handles_unused_vertices(ctx):
    pr = synth.ba_problem(n_kf=8, n_pts=300, seed=9)
    # add a keyframe and two points nobody observes: they must come back untouched
    R = np.concatenate([pr["R"], np.eye(3)[None]]); t = np.concatenate([pr["t"], [[9.0, 9.0, 9.0]]])
    fixed = np.concatenate([pr["fixed"], [0]]).astype(np.uint8)
    pts = np.concatenate([pr["pts"], [[5.0, 5.0, 5.0], [6.0, 6.0, 6.0]]])
    outs = []
    for _ in range(2):
        s = ba.LocalBA(ctx, pr["K"])
        s.set_problem(R, t, fixed, pts, pr["ecam"], pr["ept"], pr["meas"])
        s.optimize(5)
        outs.append((s.poses(), s.points(), s.chi2()))
        s.close()
    (Ra, ta), pa, ca = outs[0]; (Rb, tb), pb, cb = outs[1]
    assert np.array_equal(Ra, Rb) and np.array_equal(ta, tb) and np.array_equal(pa, pb) and np.array_equal(ca, cb)
    assert np.array_equal(pa[-2:], pts[-2:]) and np.array_equal(ta[-1], t[-1])
    o = oracle.ba_optimize(R, t, fixed, pts, pr["ecam"], pr["ept"], pr["meas"], pr["K"], 5, ba.HUBER_DELTA)
    assert _rel(pa, o["pts"]) < RTOL and _rel(ta, o["t"]) < RTOL


def test_local_ba_config3_full_size(ctx):
    """BASELINE configs[2]: 50 keyframes x 10 k landmarks x ~80 k observations, 10 LM iterations."""
    pr = synth.ba_problem()
    assert len(pr["ecam"]) > 70000
    o = oracle.ba_optimize(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"], pr["K"], 10, ba.HUBER_DELTA)
    s = ba.LocalBA(ctx, pr["K"])
    s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    done = s.optimize(10)
    R, t = s.poses(); pts = s.points()
    assert done == o["iters_done"] == 10
    assert _rel(R, o["R"]) < RTOL and _rel(t, o["t"]) < RTOL and _rel(pts, o["pts"]) < RTOL
    assert np.abs(t - pr["tgt"]).max() < np.abs(pr["t"] - pr["tgt"]).max()          # it does optimise
    s.close()


def test_bundle_adjustment_shim_mirrors_reference_api(ctx):
    pr = synth.ba_problem_frontal(n_kf=6, n_pts=120, seed=21)

    class KF:
        def __init__(self, i): self.i = i
        def idf(self): return 10 + self.i
        def Rf(self): return pr["R"][self.i]
        def tf(self): return pr["t"][self.i].reshape(3, 1)          # column vector flavour (optimizer.py:23-24)

    class MP:
        def __init__(self, j): self.j = j; self.pt3d = pr["pts"][j]
        def idf(self): return 100 + self.j
        def pt3df(self): return pr["pts"][self.j]

    class Cam:
        fx, fy, cx, cy, baseline = synth.FX, synth.FY, synth.CX, synth.CY, synth.BASELINE
    gate_cam = optimizer.CameraParameters(synth.FX, (synth.CX, synth.CY), synth.BASELINE)
    opt = optimizer.BundleAdjustment(gate_cam, ctx=ctx)
    kfs = [KF(i) for i in range(6)]; mps = {}
    for i, kf in enumerate(kfs):
        opt.add_pose(kf, Cam, fixed=bool(pr["fixed"][i]))
    kept = []
    for e, (ci, pj) in enumerate(zip(pr["ecam"], pr["ept"])):
        if pj not in mps:
            mps[pj] = MP(pj); opt.add_point(mps[pj])
        before = len(opt._edges)
        opt.add_edge(mps[pj], kfs[ci], pr["meas"][e])
        if len(opt._edges) > before:
            kept.append(e)
    assert len(pr["ecam"]) // 2 < len(kept) <= len(pr["ecam"])
    opt.optimize(7)
    kept = np.array(kept)
    # the 640x480 admission gate treats (R,t) as world->camera (optimizer.py:52-54): the oracle gets the admitted edges
    o = oracle.ba_optimize(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"][kept], pr["ept"][kept], pr["meas"][kept],
                           pr["K"], 7, ba.HUBER_DELTA)
    for i, kf in enumerate(kfs):
        R, t = opt.get_pose(kf.idf())
        assert _rel(R, o["R"][i]) < RTOL and _rel(t, o["t"][i]) < RTOL
    used = sorted(set(pr["ept"][kept]))
    for pj in used[:20]:
        assert _rel(opt.get_point(100 + pj), o["pts"][pj]) < RTOL
    bad = {(10 + pr["ecam"][e], 100 + pr["ept"][e]) for e, c in zip(kept, o["chi2"]) if c > 7.815}
    assert set(opt.outlier_edges()) == bad


def test_banded_ba_matches_dense_oracle(ctx):
    """Sequence-shaped problem (block-banded reduced camera system): the banded HIP solver vs the dense CPU oracle."""
    pr = synth.ba_problem_sequence(n_kf=150, n_pts=4000, window=6, seed=3)
    o = oracle.ba_optimize(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"], pr["K"], 6, ba.HUBER_DELTA)
    s = ba.LocalBA(ctx, pr["K"])
    s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    done = s.optimize(6)
    R, t = s.poses(); pts = s.points(); lg = s.log()[:done]
    assert done == o["iters_done"] and np.array_equal(lg[:, 2:], o["log"][:, 2:])
    assert _rel(R, o["R"]) < RTOL and _rel(t, o["t"]) < RTOL and _rel(pts, o["pts"]) < RTOL
    assert np.allclose(lg[:, 0], o["log"][:, 0], rtol=1e-8)
    s.close()


def test_global_ba_scale_properties(ctx):
    """A slice of BASELINE configs[4] (1 000 keyframes x 100 k landmarks x ~800 k observations, band +-40): too big for the
    dense oracle, checked through size-independent properties: monotone robust chi2, every step accepted, errors shrink,
    bitwise reproducibility."""
    pr = synth.ba_problem_sequence(n_kf=1000, n_pts=100000, window=20, seed=5)
    outs = []
    for _ in range(2):
        s = ba.LocalBA(ctx, pr["K"])
        s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
        done = s.optimize(5)
        lg = s.log()[:done]
        outs.append((s.poses()[1].copy(), s.points().copy(), lg.copy()))
        s.close()
    t, pts, lg = outs[0]
    assert len(lg) == 5 and (lg[:, 3] == 1).all() and np.all(np.diff(lg[:, 0]) < 0)
    assert np.array_equal(t, outs[1][0]) and np.array_equal(pts, outs[1][1])
    assert np.abs(t - pr["tgt"]).mean() < 0.5 * np.abs(pr["t"] - pr["tgt"]).mean()
    assert np.abs(pts - pr["Xgt"]).mean() < np.abs(pr["pts"] - pr["Xgt"]).mean()


def test_bcr_inside_lm_matches_dense_oracle(ctx, monkeypatch):
    """The production solver of long banded systems - block cyclic reduction, selected for >= 2 048 unknowns - inside the
    Levenberg loop against the DENSE CPU oracle: 400 keyframes (398 free, n = 2 388), once with a narrow band (window 6) and once
    with the widest one BCR takes (window 20: bw = 245).  LM log equal, states within 1e-4; and the same problem through the
    two-way window kernels (BH_BA_CHOL_BCR=0) takes the same decisions."""
    for window, seed in ((6, 3), (20, 21)):
        pr = synth.ba_problem_sequence(n_kf=400, n_pts=12000, window=window, seed=seed)
        o = oracle.ba_optimize(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"], pr["K"], 5, ba.HUBER_DELTA)
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("BH_BA_CHOL_BCR", mode)
            s = ba.LocalBA(ctx, pr["K"])
            s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
            assert s.solver() in (("bcr",) if mode == "1" else ("window", "twoway")), s.solver()
            done = s.optimize(5)
            res[mode] = (s.poses(), s.points().copy(), s.log()[:done].copy(), done)
            s.close()
        monkeypatch.delenv("BH_BA_CHOL_BCR")
        for mode in ("1", "0"):
            (R, t), pts, lg, done = res[mode]
            assert done == o["iters_done"] and np.array_equal(lg[:, 2:], o["log"][:, 2:])
            assert np.allclose(lg[:, 0], o["log"][:, 0], rtol=1e-8)
            assert _rel(R, o["R"]) < RTOL and _rel(t, o["t"]) < RTOL and _rel(pts, o["pts"]) < RTOL


def test_global_ba_config5_full_size(ctx, monkeypatch):
    """BASELINE configs[4] at its full size (5 000 keyframes, 500 k landmarks, 3.9 M observations): too big for the dense oracle;
    the two solvers of the reduced camera system (block cyclic reduction over the whole chip = the default; two-way window
    elimination = BH_BA_CHOL_BCR=0) must agree, chi2 must fall monotonically with every step accepted, and the estimate must
    move towards the ground truth."""
    pr = synth.ba_problem_sequence()
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BH_BA_CHOL_BCR", mode)
        s = ba.LocalBA(ctx, pr["K"])
        s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
        assert s.solver() in (("bcr",) if mode == "1" else ("window", "twoway")), s.solver()
        done = s.optimize(4)
        res[mode] = (s.poses()[1].copy(), s.points().copy(), s.log()[:done].copy())
        s.close()
    monkeypatch.delenv("BH_BA_CHOL_BCR")
    t, pts, lg = res["1"]
    assert len(lg) == 4 and (lg[:, 3] == 1).all() and np.all(np.diff(lg[:, 0]) < 0)
    assert np.array_equal(lg[:, 2:], res["0"][2][:, 2:]) and np.allclose(lg[:, 0], res["0"][2][:, 0], rtol=1e-10)
    assert np.abs(t - res["0"][0]).max() < 1e-8 and np.abs(pts - res["0"][1]).max() < 1e-8
    assert np.abs(t - pr["tgt"]).mean() < 0.2 * np.abs(pr["t"] - pr["tgt"]).mean()


def test_device_built_problem_equals_host_built(ctx, monkeypatch):
    """bh_ba_set_problem builds the graph structure (slots, edges by point / by camera, camera pairs, join lists, band) on the device:
    local windows (<= 64 free keyframes) by ONE workgroup, everything else by chip-wide kernels (BH_BA_BUILD=general: counting, scans and
    ranking, no sort; its camera pairs either from band counters or - BH_BA_PAIRS=bitmap, the wide-band path - from a Cf x Cf bitmap);
    the former host loops stay selectable (BH_BA_BUILD=host).  Same problem through all four: every structural array identical, and
    poses, landmarks, per-edge chi2 and the LM log equal bit for bit."""
    probs = [synth.ba_problem(n_kf=12, n_pts=500, seed=4), synth.ba_problem(n_kf=50, n_pts=10000, seed=7),
             synth.ba_problem_sequence(n_kf=300, n_pts=9000, window=6, seed=3), synth.ba_problem(n_kf=3, n_pts=40, seed=11),
             synth.ba_problem(n_kf=20, n_pts=1500, seed=9)]
    pr = synth.ba_problem(n_kf=9, n_pts=200, seed=2)                       # unused point, unused camera, fixed camera with edges
    keep = (pr["ecam"] != 4) & (pr["ept"] != 17)
    probs.append(dict(pr, ecam=pr["ecam"][keep], ept=pr["ept"][keep], meas=pr["meas"][keep]))
    pr = synth.ba_problem(n_kf=14, n_pts=300, seed=6)                      # duplicate observations (ties keep the input position), shuffled input
    rng = np.random.default_rng(1)
    dup = rng.choice(len(pr["ecam"]), 150, replace=False)
    ecam = np.concatenate([pr["ecam"], pr["ecam"][dup]]); ept = np.concatenate([pr["ept"], pr["ept"][dup]])
    meas = np.concatenate([pr["meas"], pr["meas"][dup] + rng.normal(0, 0.3, (150, 2))])
    sh = rng.permutation(len(ecam))
    probs.append(dict(pr, ecam=np.ascontiguousarray(ecam[sh]), ept=np.ascontiguousarray(ept[sh]), meas=np.ascontiguousarray(meas[sh])))
    pr = synth.ba_problem(n_kf=40, n_pts=2500, seed=8)                     # a window with most keyframes fixed (the outer ring of covisibility_graph.py:200-208)
    fx = pr["fixed"].copy(); fx[8:] = 1
    probs.append(dict(pr, fixed=fx))
    for pi, pr in enumerate(probs):
        res, st = [], []
        for mode in ("device", "general", "host", "bitmap"):
            monkeypatch.delenv("BH_BA_PAIRS", raising=False)
            if mode == "device":
                monkeypatch.delenv("BH_BA_BUILD", raising=False)
            elif mode == "bitmap":
                monkeypatch.setenv("BH_BA_BUILD", "general"); monkeypatch.setenv("BH_BA_PAIRS", "bitmap")
            else:
                monkeypatch.setenv("BH_BA_BUILD", mode)
            s = ba.LocalBA(ctx, pr["K"])
            s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
            st.append(s.structure())
            done = s.optimize(4)
            res.append((s.poses(), s.points(), s.chi2(), s.log()[:done]))
            s.close()
        monkeypatch.delenv("BH_BA_BUILD", raising=False); monkeypatch.delenv("BH_BA_PAIRS", raising=False)
        for other in (1, 2, 3):
            for name in st[0]:
                assert np.array_equal(st[0][name], st[other][name]), (pi, name, other)
            (Rd, td), pd, cd, ld = res[0]; (Rh, th), ph, ch, lh = res[other]
            assert np.array_equal(ld, lh) and np.array_equal(Rd, Rh) and np.array_equal(td, th)
            assert np.array_equal(pd, ph, equal_nan=True) and np.array_equal(cd, ch)


def test_batched_windows_equal_single_window_runs_bitwise(ctx, monkeypatch):
    """bh_ba_batch_optimize: W windows of different sizes advance in lockstep (one launch per kernel and trial for all of them).  Every
    window must come out bit for bit as bh_ba_optimize leaves it: poses, landmarks, per-edge chi2, the Levenberg log - including a
    window that needs retries (a gross outlier start), one with no free landmark edges removed, and non-verbose error semantics.
    The single-window reference is the HOST-driven loop of bh_ba_optimize (BH_BA_SELF_BATCH=0: one trial enqueued at a time, the host
    deciding between linearisation and retry from the verdict it reads); by default a local window runs as a batch of one with the control
    flow on the device and the host one round ahead - that default is compared with the host loop too."""
    monkeypatch.setenv("BH_BA_SELF_BATCH", "0")
    probs = [synth.ba_problem(n_kf=6, n_pts=150, seed=1), synth.ba_problem(n_kf=12, n_pts=500, seed=4), synth.ba_problem(n_kf=50, n_pts=10000, seed=7),
             synth.ba_problem(n_kf=20, n_pts=1500, seed=9), synth.ba_problem(n_kf=9, n_pts=200, seed=2), synth.ba_problem(n_kf=30, n_pts=4000, seed=12)]
    for seed in (9, 14):                                                    # small windows that reach the numerical floor: rejected trials, lambda grows
        bad = synth.ba_problem(n_kf=5, n_pts=60, seed=seed)
        rng = np.random.default_rng(seed)
        probs.append(dict(bad, t=bad["t"] + rng.normal(0, 0.1, bad["t"].shape) * (1 - bad["fixed"])[:, None], pts=bad["pts"] + rng.normal(0, 0.3, bad["pts"].shape)))
    NIT = 10
    for verbose in (True, False):
        single = []
        for pr in probs:
            s = ba.LocalBA(ctx, pr["K"])
            s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
            done = s.optimize(NIT, verbose_errors=verbose)
            single.append((done, s.poses(), s.points().copy(), s.chi2().copy(), s.log()[:done].copy()))
            s.close()
        monkeypatch.delenv("BH_BA_SELF_BATCH")
        for pr, ref in list(zip(probs, single))[::3]:                        # the default path of bh_ba_optimize: a batch of one, host one round ahead
            s = ba.LocalBA(ctx, pr["K"])
            s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
            done = s.optimize(NIT, verbose_errors=verbose)
            R, t = s.poses()
            assert done == ref[0] and np.array_equal(R, ref[1][0]) and np.array_equal(t, ref[1][1]) and np.array_equal(s.points(), ref[2], equal_nan=True)
            assert np.array_equal(s.chi2(), ref[3]) and np.array_equal(s.log()[:done], ref[4])
            s.close()
        monkeypatch.setenv("BH_BA_SELF_BATCH", "0")
        wins = [ba.LocalBA(ctx, pr["K"]) for pr in probs]
        # all windows set up by ONE call: the local ones are built together (a workgroup each), the 50-keyframe one goes the general way
        ba.set_problems(ctx, wins, [(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"]) for pr in probs])
        if verbose:
            for pr, s in zip(probs, wins):
                one = ba.LocalBA(ctx, pr["K"])
                one.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
                a, b = one.structure(), s.structure()
                assert all(np.array_equal(a[k], b[k]) for k in a)
                one.close()
        batch = ba.LocalBABatch(ctx, wins)
        done = batch.optimize(NIT, verbose_errors=verbose)
        assert any(sg[4][:, 2].max() > 1 for sg in single), "no window exercised the retry path"
        for w, s in enumerate(wins):
            d0, (R0, t0), p0, c0, l0 = single[w]
            assert done[w] == d0
            R, t = s.poses()
            assert np.array_equal(R, R0) and np.array_equal(t, t0) and np.array_equal(s.points(), p0, equal_nan=True)
            assert np.array_equal(s.chi2(), c0) and np.array_equal(s.log()[:d0], l0)
        if verbose:                                                         # a window count that is not a multiple of 8 (the grids are padded to 8 windows)
            three = [ba.LocalBA(ctx, pr["K"]) for pr in probs[:3]]
            ba.set_problems(ctx, three, [(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"]) for pr in probs[:3]])
            b3 = ba.LocalBABatch(ctx, three)
            d3 = b3.optimize(NIT, verbose_errors=verbose)
            for w, s3 in enumerate(three):
                assert d3[w] == single[w][0] and np.array_equal(s3.poses()[1], single[w][1][1]) and np.array_equal(s3.points(), single[w][2], equal_nan=True)
            b3.close()
            for s3 in three:
                s3.close()
        # a second run on the same handles continues from the optimised state exactly like the single-window path does
        done2 = batch.optimize(2, verbose_errors=verbose)
        assert (done2 >= 1).all()
        batch.close()
        for s in wins:
            s.close()
    # a window the batch does not take (a long banded problem) is refused with a message, not run
    pr = synth.ba_problem_sequence(n_kf=150, n_pts=4000, window=6, seed=3)
    s = ba.LocalBA(ctx, pr["K"])
    s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    batch = ba.LocalBABatch(ctx, [s])
    with pytest.raises(Exception, match="local windows"):
        batch.optimize(2)
    batch.close(); s.close()


def test_set_problems_mixed_windows_and_errors(ctx):
    """bh_ba_batch_set_problems with every kind of window in one call: a live window (one-workgroup builder, LDS), a configs[2] window
    (the same steps as grid-wide kernels), a 300-keyframe sequence (chip-wide builder: more than 64 free keyframes) and an empty one;
    each must come out as bh_ba_set_problem builds it.  A bad edge in any window fails the call and leaves no window half-built."""
    probs = [synth.ba_problem(n_kf=10, n_pts=600, seed=21), synth.ba_problem(n_kf=50, n_pts=10000, seed=7),
             synth.ba_problem_sequence(n_kf=300, n_pts=9000, window=6, seed=3), synth.ba_problem(n_kf=4, n_pts=30, seed=5)]
    args = [(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"]) for pr in probs]
    args.append((probs[3]["R"], probs[3]["t"], probs[3]["fixed"], probs[3]["pts"], probs[3]["ecam"][:0], probs[3]["ept"][:0], probs[3]["meas"][:0]))
    wins = [ba.LocalBA(ctx, probs[0]["K"]) for _ in args]
    ba.set_problems(ctx, wins, args)
    for a, s in zip(args, wins):
        one = ba.LocalBA(ctx, probs[0]["K"])
        one.set_problem(*a)
        sa, sb = one.structure(), s.structure()
        assert all(np.array_equal(sa[k], sb[k]) for k in sa)
        d1, d2 = one.optimize(3), s.optimize(3)
        assert d1 == d2 and np.array_equal(one.log()[:d1], s.log()[:d2]) and np.array_equal(one.poses()[1], s.poses()[1])
        one.close()
    bad = list(args[0]); e = bad[4].copy(); e[17] = 99; bad[4] = e
    with pytest.raises(RuntimeError, match="edge 17"):
        ba.set_problems(ctx, wins[:2], [args[1], tuple(bad)])
    with pytest.raises(RuntimeError, match="no problem"):
        wins[0].optimize(1)
    for s in wins:
        s.close()


def test_control_sums_and_lookahead_switches_agree(ctx, monkeypatch):
    """Round 5: the sums of the Levenberg control ride in the producing kernels (fan-in: per-workgroup slots, the last arriver adds them up) and
    the host enqueues a round before it has seen the last one's verdicts.  Both have an A/B switch: BH_BA_LOOKAHEAD=0 must not change a bit;
    BH_BA_BB_TAIL=0 (a batched window's control steps as launches of their own instead of in the last workgroup to arrive) neither, nor
    BH_BA_BUILD_VALUES_FIRST=1 (the problem build of the 50-keyframe window sends landmarks / measurements / camera blocks on the context's stream in front
    of the structure kernels instead of on the auxiliary stream beside them);
    BH_BA_FAN=0 (the sums as separate launches, another fixed summation order) must give the same Levenberg decisions and the same states to
    rounding - on a local window (host loop), and on the banded solver path."""
    for pr, env in ((synth.ba_problem(n_kf=12, n_pts=500, seed=4), {}), (synth.ba_problem(n_kf=50, n_pts=10000, seed=7), {}),
                    (synth.ba_problem(n_kf=12, n_pts=500, seed=4), {"BH_BA_SELF_BATCH": "0"})):
        res = {}
        for name, sw in (("default", {}), ("nolook", {"BH_BA_LOOKAHEAD": "0"}), ("notail", {"BH_BA_BB_TAIL": "0"}), ("valuesfirst", {"BH_BA_BUILD_VALUES_FIRST": "1"}),
                         ("nofan", {"BH_BA_FAN": "0", "BH_BA_SELF_BATCH": "0"}),
                         ("oneper", {"BH_BA_CAM_PARTS": "1"}), ("eightper", {"BH_BA_CAM_PARTS": "8"})):
            for k, v in {**env, **sw}.items():
                monkeypatch.setenv(k, v)
            s = ba.LocalBA(ctx, pr["K"])
            s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
            done = s.optimize(10)
            res[name] = (done, s.poses(), s.points().copy(), s.log()[:done].copy())
            s.close()
            for k in {**env, **sw}:
                monkeypatch.delenv(k)
        d0, (R0, t0), p0, l0 = res["default"]
        for name in ("nolook", "notail", "valuesfirst"):      # who enqueues when, which workgroup adds the slots up, and on which stream a big window's values go up, must not change a bit
            d1, (R1, t1), p1, l1 = res[name]
            assert d1 == d0 and np.array_equal(R1, R0) and np.array_equal(t1, t0) and np.array_equal(p1, p0) and np.array_equal(l1, l0), name
        # another fixed summation order: BH_BA_FAN=0, and the camera blocks of the linearisation summed by one / eight workgroups per keyframe
        # (BH_BA_CAM_PARTS; the default cuts the 1 600-observation lists of the 50-keyframe window into parts, the 12-keyframe window's not)
        for name in ("nofan", "oneper", "eightper"):
            d2, (R2, t2), p2, l2 = res[name]
            assert d2 == d0 and np.array_equal(l2[:, 2:], l0[:, 2:]), name                # trials per iteration, accept flags
            assert np.allclose(l2[:, :2], l0[:, :2], rtol=1e-9) and np.abs(R2 - R0).max() < 1e-9 and np.abs(t2 - t0).max() < 1e-9 and np.abs(p2 - p0).max() < 1e-9, name
