"""GPU parity on edge cases: odd image sizes (unaligned width path), few levels, blank frames, tiny budgets, capacity errors,
and the committed golden vectors."""
import os
import numpy as np
import pytest

import oracle
from boslam_amd import BhError, frontend, synth, cv2_shim, ba, optimizer

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _check(ctx, imgs, mask, params_kw, max_kp=None):
    params = frontend.make_params(**params_kw)
    kps, desc, n = frontend.orb_detect_describe(ctx, imgs, mask=mask, params=params, max_kp=max_kp)
    P = oracle.orb_params(**params_kw)
    for f in range(len(imgs)):
        ok, od, rc = oracle.orb_detect_describe(imgs[f], None if mask is None else mask[f], P, max_kp=4 * params_kw.get("nfeatures", 600) + 256)
        assert rc == 0
        assert n[f] == len(ok), (f, n[f], len(ok))
        assert np.array_equal(desc[f, :n[f]], od)
        for fld in ("x", "y", "angle", "response", "size"):
            assert np.array_equal(kps[f, :n[f]][fld].view(np.uint32), ok[fld].view(np.uint32)), fld
        assert np.array_equal(kps[f, :n[f]]["octave"], ok["octave"])
    return n


@pytest.mark.parametrize("H,W", [(241, 322), (200, 203), (130, 161), (480, 636)])
def test_odd_sizes_gray_and_bgr(ctx, H, W):
    rng = np.random.default_rng(H * 1000 + W)
    bgr, _ = synth.render_batch(2)
    oy, ox = min(7, 480 - H), min(11, 640 - W)
    crop = np.ascontiguousarray(bgr[:, oy:oy + H, ox:ox + W])
    mask = (rng.random((2, H, W)) > 0.1).astype(np.uint8) * 255
    n = _check(ctx, crop, mask, dict(nfeatures=300))
    assert n.sum() > 0
    _check(ctx, np.ascontiguousarray(crop[..., 1]), None, dict(nfeatures=200))


@pytest.mark.parametrize("H,W", [(241, 322), (200, 203), (130, 161), (480, 636), (97, 129)])
def test_depth_masks_take_the_bit_packed_pyramid(ctx, H, W):
    """mask = (depth > 0) * 255 (camera.py:141) is binary, so the device keeps it one bit per pixel: salt-and-pepper holes (every
    resize window mixed), rectangular holes (long uniform runs with edges), widths that are not multiples of 4 / 32 / 64."""
    rng = np.random.default_rng(H * 7 + W)
    bgr, _ = synth.render_batch(2)
    oy, ox = min(7, 480 - H), min(11, 640 - W)
    crop = np.ascontiguousarray(bgr[:, oy:oy + H, ox:ox + W])
    depth = rng.integers(1, 5000, size=(2, H, W)).astype(np.uint16)
    depth[0][rng.random((H, W)) < 0.15] = 0
    for _ in range(12):
        y, x, hh, ww = rng.integers(0, H - 8), rng.integers(0, W - 8), rng.integers(4, 60), rng.integers(4, 90)
        depth[1, y:y + hh, x:x + ww] = 0
    params_kw = dict(nfeatures=300)
    params = frontend.make_params(**params_kw)
    kps, desc, n = frontend.orb_detect_describe(ctx, crop, depth=depth, params=params)
    mask = (depth > 0).astype(np.uint8) * 255
    for f in range(2):
        ok, od, rc, dbg = oracle.orb_detect_describe(crop[f], mask[f], oracle.orb_params(**params_kw), max_kp=4 * 300 + 256, debug=True)
        dev = frontend.orb_debug_fetch(ctx, f, H, W, params)              # the mask pyramid itself, expanded from the bits
        for l in range(params.nlevels):
            w, h, p, o = int(dev["w"][l]), int(dev["h"][l]), int(dev["pitch"][l]), int(dev["off"][l])
            got = dev["mask"][o:o + p * h].reshape(h, p)[:, :w]
            ro = int(dbg["off"][l])
            bad = np.argwhere(got != dbg["mask"][ro:ro + w * h].reshape(h, w))
            assert len(bad) == 0, (f, l, len(bad), bad[:3])
        assert rc == 0 and n[f] == len(ok), (f, n[f], len(ok))
        assert np.array_equal(desc[f, :n[f]], od)
        for fld in ("x", "y", "angle", "response", "size"):
            assert np.array_equal(kps[f, :n[f]][fld].view(np.uint32), ok[fld].view(np.uint32)), fld
    assert n.sum() > 0


@pytest.mark.parametrize("H,W,scale,nlevels", [(480, 640, 1.2, 8), (123, 200, 1.2, 8), (250, 324, 1.3, 4), (96, 128, 1.25, 3), (301, 404, 1.07, 6)])
def test_level0_and_level1_in_one_launch(ctx, monkeypatch, H, W, scale, nlevels):
    """Round 5: k_gray_l1 makes the gray + mask-bit level 0 and level 1 of the pyramid in one launch (strips of level-1 rows, their level-0
    rows kept in LDS) when the frame is BGR + depth with a width that is a multiple of 4.  Strip counts that do not divide the level, other
    scale factors in the supported range (other strip heights), short pyramids: every level of the image and of the mask pyramid and everything downstream must
    equal the oracle, and the two-launch form (BH_ORB_GRAY_L1=0) must give the same bytes.  BH_ORB_PYR_PAIR=1 makes the later levels in pairs the
    same way (k_pyr_pair: levels 2 + 3, 4 + 5, 6 + 7 behind k_gray_l1 - form "pairs"; 1 + 2, 3 + 4, ... in the two-launch form - form "two"; an odd
    level out by k_pyr_down_bits): same bytes again."""
    rng = np.random.default_rng(H + 3 * W)
    bgr, _ = synth.render_batch(2)
    crop = np.ascontiguousarray(bgr[:, 5:5 + H, 9:9 + W]) if H < 470 else np.ascontiguousarray(bgr[:, :H, :W])
    depth = rng.integers(1, 5000, size=(2, H, W)).astype(np.uint16)
    depth[0][rng.random((H, W)) < 0.2] = 0
    for _ in range(10):
        y, x, hh, ww = rng.integers(0, H - 8), rng.integers(0, W - 8), rng.integers(4, 50), rng.integers(4, 80)
        depth[1, y:y + hh, x:x + ww] = 0
    kw = dict(nfeatures=300, scaleFactor=scale, nlevels=nlevels)
    params = frontend.make_params(**kw)
    mask = (depth > 0).astype(np.uint8) * 255
    out = {}
    for form in ("fused", "two", "pairs"):
        if form in ("two", "pairs"):
            monkeypatch.setenv("BH_ORB_PYR_PAIR", "1")             # "two": levels 1 + 2, 3 + 4, ... in pairs; "pairs": 2 + 3, 4 + 5, ... behind k_gray_l1
        if form == "two":
            monkeypatch.setenv("BH_ORB_GRAY_L1", "0")
        kps, desc, n = frontend.orb_detect_describe(ctx, crop, depth=depth, params=params)
        lv = []
        for f in range(2):
            dev = frontend.orb_debug_fetch(ctx, f, H, W, params)
            lv.append({name: [dev[name][int(dev["off"][l]):int(dev["off"][l]) + int(dev["pitch"][l]) * int(dev["h"][l])].reshape(int(dev["h"][l]), int(dev["pitch"][l]))[:, :int(dev["w"][l])].copy()
                              for l in range(nlevels)] for name in ("pyr", "mask")})
        out[form] = (kps.copy(), desc.copy(), n.copy(), lv)
        if form == "two":
            monkeypatch.delenv("BH_ORB_GRAY_L1")
        if form in ("two", "pairs"):
            monkeypatch.delenv("BH_ORB_PYR_PAIR")
    kps, desc, n, lv = out["fused"]
    k2, d2, n2, lv2 = out["two"]
    k3, d3, n3, lv3 = out["pairs"]
    assert np.array_equal(n, n2) and np.array_equal(n, n3)
    assert np.array_equal(d3, desc) and np.array_equal(k3.view(np.uint8), kps.view(np.uint8))
    for f in range(2):
        ok, od, rc, dbg = oracle.orb_detect_describe(crop[f], mask[f], oracle.orb_params(**kw), max_kp=4 * 300 + 256, debug=True)
        for name in ("pyr", "mask"):
            for l in range(nlevels):
                w, h, ro = int(dbg["w"][l]), int(dbg["h"][l]), int(dbg["off"][l])
                ref = dbg[name][ro:ro + w * h].reshape(h, w)
                bad = np.argwhere(lv[f][name][l] != ref)
                assert len(bad) == 0, (name, f, l, len(bad), bad[:3])
                assert np.array_equal(lv2[f][name][l], ref), (name, f, l, "two-launch form")
                assert np.array_equal(lv3[f][name][l], ref), (name, f, l, "levels in pairs")
        assert rc == 0 and n[f] == len(ok), (f, n[f], len(ok))
        assert np.array_equal(desc[f, :n[f]], od) and np.array_equal(d2[f, :n[f]], od)
        for fld in ("x", "y", "angle", "response", "size"):
            assert np.array_equal(kps[f, :n[f]][fld].view(np.uint32), ok[fld].view(np.uint32)), fld
            assert np.array_equal(k2[f, :n[f]][fld].view(np.uint32), ok[fld].view(np.uint32)), fld


@pytest.mark.parametrize("H,W", [(480, 640), (241, 322), (97, 129)])
def test_matrix_core_pyramid_is_bit_identical(monkeypatch, H, W):
    """BH_ORB_PYR_MFMA=1 selects k_pyr_mfma (resize as two int8 matrix-core products, mask by run gathers + truth-table patches) instead of
    k_pyr_down_bits: every level of the image and of the mask pyramid, and everything downstream, must equal the oracle all the same."""
    from boslam_amd import Context
    monkeypatch.setenv("BH_ORB_PYR_MFMA", "1")
    c = Context(0)                                               # the choice is made when a context builds its geometry tables
    try:
        rng = np.random.default_rng(H * 3 + W)
        bgr, _ = synth.render_batch(2)
        crop = np.ascontiguousarray(bgr[:, :H, :W])
        depth = rng.integers(1, 5000, size=(2, H, W)).astype(np.uint16)
        depth[0][rng.random((H, W)) < 0.15] = 0
        for _ in range(12):
            y, x, hh, ww = rng.integers(0, H - 8), rng.integers(0, W - 8), rng.integers(4, 60), rng.integers(4, 90)
            depth[1, y:y + hh, x:x + ww] = 0
        params = frontend.make_params(nfeatures=400)
        kps, desc, n = frontend.orb_detect_describe(c, crop, depth=depth, params=params)
        mask = (depth > 0).astype(np.uint8) * 255
        for f in range(2):
            ok, od, rc, dbg = oracle.orb_detect_describe(crop[f], mask[f], oracle.orb_params(nfeatures=400), max_kp=4 * 400 + 256, debug=True)
            dev = frontend.orb_debug_fetch(c, f, H, W, params)
            for name in ("pyr", "mask"):
                for l in range(params.nlevels):
                    w, h, p, o = int(dev["w"][l]), int(dev["h"][l]), int(dev["pitch"][l]), int(dev["off"][l])
                    got = dev[name][o:o + p * h].reshape(h, p)[:, :w]
                    ro = int(dbg["off"][l])
                    bad = np.argwhere(got != dbg[name][ro:ro + w * h].reshape(h, w))
                    assert len(bad) == 0, (name, f, l, len(bad), bad[:3])
            assert rc == 0 and n[f] == len(ok)
            assert np.array_equal(desc[f, :n[f]], od)
    finally:
        c.close()


@pytest.mark.parametrize("nlevels", [1, 3, 12])
def test_depth_masks_at_other_pyramid_depths(ctx, nlevels):
    """The bit-packed mask path with no resize at all (one level), a short pyramid, and levels smaller than the border."""
    rng = np.random.default_rng(nlevels)
    bgr, depth = synth.render_batch(2)
    depth = depth.copy()
    depth[0][rng.random(depth[0].shape) < 0.2] = 0
    depth[1, 200:300, 100:400] = 0
    kw = dict(nfeatures=400, nlevels=nlevels)
    params = frontend.make_params(**kw)
    kps, desc, n = frontend.orb_detect_describe(ctx, bgr, depth=depth, params=params)
    mask = (depth > 0).astype(np.uint8) * 255
    for f in range(2):
        ok, od, rc = oracle.orb_detect_describe(bgr[f], mask[f], oracle.orb_params(**kw), max_kp=4 * 400 + 256)
        assert rc == 0 and n[f] == len(ok), (f, n[f], len(ok))
        assert np.array_equal(desc[f, :n[f]], od)
        for fld in ("x", "y", "angle", "response", "size"):
            assert np.array_equal(kps[f, :n[f]][fld].view(np.uint32), ok[fld].view(np.uint32)), fld


def test_non_binary_mask_bytes_keep_the_byte_pyramid(ctx):
    """A caller-supplied mask may hold any byte: level 0 filters on != 0, the levels above are resize + threshold(254) of the bytes
    (a 254 next to 255s can still round to 255).  That path keeps the byte mask pyramid; only depth-derived masks are bit-packed."""
    rng = np.random.default_rng(123)
    bgr, _ = synth.render_batch(2)
    mask = rng.choice(np.array([0, 1, 128, 254, 255], np.uint8), size=(2, 480, 640), p=[0.1, 0.05, 0.05, 0.2, 0.6])
    mask[1, 100:300, 200:500] = 255
    n = _check(ctx, bgr, mask, dict(nfeatures=500))
    assert n.sum() > 0


def test_noise_tiles_take_the_list_free_path(ctx):
    """White noise with a low FAST threshold: well over half of the pixels pass the detector's pretest, more than k_fast's candidate
    list holds, so its tiles score every pixel with the full ring test instead - same keypoints and descriptors as the oracle.
    Half of the second frame is flat, so both kinds of tile occur in one launch."""
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, size=(2, 200, 264), dtype=np.uint8)
    img[1, :, 132:] = 90
    for thr in (5, 20):
        n = _check(ctx, img, None, dict(nfeatures=400, fastThreshold=thr))
        assert n.min() > 100


def test_single_level_and_three_levels(ctx):
    bgr, depth = synth.render_batch(2)
    mask = (depth > 0).astype(np.uint8) * 255
    _check(ctx, bgr, mask, dict(nfeatures=400, nlevels=1))
    _check(ctx, bgr, mask, dict(nfeatures=400, nlevels=3, scaleFactor=1.25, fastThreshold=30))
    _check(ctx, bgr, None, dict(nfeatures=400, nlevels=12))              # top levels smaller than the 2x31-px border: no keypoints there


def test_blank_and_fully_masked_frames(ctx):
    imgs = np.zeros((3, 480, 640, 3), np.uint8); imgs[1] = 127
    bgr, _ = synth.render_batch(1)
    imgs[2] = bgr[0]
    mask = np.full((3, 480, 640), 255, np.uint8); mask[2] = 0             # real frame, everything masked out
    n = _check(ctx, imgs, mask, dict(nfeatures=600))
    assert list(n) == [0, 0, 0]
    kp, des = cv2_shim.ORB_create(nfeatures=600, ctx=ctx).detectAndCompute(imgs[0], None)
    assert kp == [] and des is None                                       # experiments/pnp_two_way_tracking.py:38
    kp, des = cv2_shim.ORB_create(nfeatures=600, ctx=ctx).detectAndCompute(bgr[0], None)
    assert len(kp) == 600 and des.shape == (600, 32) and isinstance(kp[0].pt, tuple)


def test_tiny_budget_and_capacity_error(ctx):
    bgr, depth = synth.render_batch(1)
    mask = (depth > 0).astype(np.uint8) * 255
    _check(ctx, bgr, mask, dict(nfeatures=8))
    _check(ctx, bgr, mask, dict(nfeatures=0))
    with pytest.raises(BhError) as e:
        frontend.orb_detect_describe(ctx, bgr, mask=mask, params=frontend.make_params(nfeatures=600), max_kp=100)
    assert e.value.code == -3                                             # BH_ERR_CAPACITY
    assert len(frontend.orb_detect_describe(ctx, bgr[:0], params=frontend.make_params())[2]) == 0      # empty batch


def test_bf_matcher_shim_edge_cases(ctx):
    m = cv2_shim.BFMatcher_create(cv2_shim.NORM_HAMMING, crossCheck=True, ctx=ctx)
    d = synth.random_descriptors(1, 50)
    assert m.match(d[:0], d) == [] and m.match(d, d[:0]) == [] and m.match(None, d) == []
    r = m.match(d, d)
    assert [x.queryIdx for x in r] == list(range(50)) and all(x.distance == 0.0 and x.trainIdx == x.queryIdx for x in r)
    with pytest.raises(ValueError):
        cv2_shim.BFMatcher_create(cv2_shim.NORM_HAMMING, crossCheck=False)


def test_hip_reproduces_committed_golden_vectors(ctx):
    g = np.load(os.path.join(GOLD, "frontend_golden.npz"))
    bgr, depth = synth.render_batch(2)
    kps, desc, n = frontend.orb_detect_describe(ctx, bgr, depth=depth, params=frontend.make_params(nfeatures=600))
    for f in range(2):
        assert np.array_equal(desc[f, :n[f]], g[f"desc{f}"])
        for fld in ("x", "y", "angle", "response", "size"):
            assert np.array_equal(kps[f, :n[f]][fld].view(np.uint32), g[f"kp{f}_{fld}"].view(np.uint32)), fld
    qi, ti, d, nm = frontend.match_pairs(ctx, desc[1:2], n[1:2], desc[0:1], n[0:1])
    assert np.array_equal(np.stack([qi[0, :nm[0]], ti[0, :nm[0]], d[0, :nm[0]]], 1), g["match_1_vs_0"])
    b = np.load(os.path.join(GOLD, "ba_golden.npz"))
    p = synth.pose_problem(n=120, seed=11)
    R, t, chi2, inl, st = ba.pose_only_ba_batch(ctx, p["pix"][None], p["pts"][None], [120], p["K"])
    assert np.allclose(R[0], b["pose_R"], atol=1e-9) and np.allclose(t[0], b["pose_t"], atol=1e-9) and np.array_equal(inl[0], b["pose_inlier"])
    pr = synth.ba_problem(n_kf=6, n_pts=150, seed=1)
    s = ba.LocalBA(ctx, pr["K"]); s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"]); s.optimize(7)
    assert np.allclose(s.poses()[1], b["ba_t"], atol=1e-8) and np.allclose(s.points(), b["ba_pts"], atol=1e-8)
    s.close()


def test_ba_degenerate_inputs(ctx):
    # no edges at all: optimise is a no-op, poses/points come back unchanged
    s = ba.LocalBA(ctx, [synth.FX, synth.FY, synth.CX, synth.CY, synth.BASELINE])
    R = np.eye(3)[None].repeat(2, 0); t = np.zeros((2, 3)); pts = np.ones((3, 3))
    s.set_problem(R, t, [1, 0], pts, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2)))
    assert s.optimize(5) == 0
    assert np.array_equal(s.points(), pts) and np.allclose(s.poses()[0], R)
    s.close()
    # all keyframes fixed: only the points move (structure-only BA), must match the oracle
    pr = synth.ba_problem(n_kf=5, n_pts=80, seed=8)
    fixed = np.ones(5, np.uint8)
    o = oracle.ba_optimize(pr["R"], pr["t"], fixed, pr["pts"], pr["ecam"], pr["ept"], pr["meas"], pr["K"], 5, ba.HUBER_DELTA)
    s = ba.LocalBA(ctx, pr["K"]); s.set_problem(pr["R"], pr["t"], fixed, pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    assert s.optimize(5) == o["iters_done"]
    assert np.abs(s.points() - o["pts"]).max() < 1e-6 and np.array_equal(s.poses()[1], pr["t"])
    s.close()
    # CamOnlyBA with too few / degenerate points must not raise (reference swallows everything)
    class Cam: fx, fy, cx, cy, baseline = synth.FX, synth.FY, synth.CX, synth.CY, synth.BASELINE
    r = optimizer.CamOnlyBA(np.zeros((0, 2)), np.zeros((0, 3)), Cam, ctx=ctx)
    assert r[0] in (True, False)


def test_new_entry_points_reject_bad_arguments(ctx):
    """Error behaviour of the round-1 additions: every bad call returns a negative code with a message, nothing crashes."""
    import ctypes as C
    from boslam_amd import _lib, window
    L = _lib.lib()
    # window problem: capacity too small -> BH_ERR_CAPACITY with the needed sizes in the message
    nbr_start = np.array([0, 1, 2], np.int32); nbr_id = np.array([1, 0], np.int32); nbr_common = np.array([20, 20], np.int32)
    obs_start = np.array([0, 2, 4], np.int32); obs_mp = np.array([0, 1, 1, 0], np.int32); obs_px = np.zeros((4, 2), np.int64)
    win = np.zeros(1, np.int32); fix = np.zeros(1, np.uint8); pt = np.zeros(1, np.int32); ep = np.zeros(1, np.int32); ept = np.zeros(1, np.int32)
    epx = np.zeros((1, 2), np.int64); cnt = np.zeros(5, np.int32)
    rc = L.bh_ba_window_problem(ctx.h, 2, _lib.ptr(nbr_start), _lib.ptr(nbr_id), _lib.ptr(nbr_common), _lib.ptr(obs_start), _lib.ptr(obs_mp),
                                _lib.ptr(obs_px), 2, 0, 15, 1, _lib.ptr(win), _lib.ptr(fix), 1, _lib.ptr(pt), 1, _lib.ptr(ep), _lib.ptr(ept),
                                _lib.ptr(epx), _lib.ptr(cnt))
    assert rc == -3 and b"capacity" in L.bh_last_error(ctx.h)
    assert cnt[0] == 2 and cnt[1] == 1                                    # the needed window size is still reported
    # ... and with room it is the two keyframes, two shared points, four edges
    w = window.window_problem(ctx, nbr_start, nbr_id, nbr_common, obs_start, obs_mp, obs_px, n_mp=2, kf=0)
    assert list(w["win_kf"]) == [0, 1] and list(w["win_fixed"]) == [True, False] and len(w["edge_pose"]) == 4
    # out-of-range ids
    bad = obs_mp.copy(); bad[0] = 7
    rc = L.bh_ba_window_problem(ctx.h, 2, _lib.ptr(nbr_start), _lib.ptr(nbr_id), _lib.ptr(nbr_common), _lib.ptr(obs_start), _lib.ptr(bad),
                                _lib.ptr(obs_px), 2, 0, 15, 4, _lib.ptr(np.zeros(4, np.int32)), _lib.ptr(np.zeros(4, np.uint8)), 4,
                                _lib.ptr(np.zeros(4, np.int32)), 4, _lib.ptr(np.zeros(4, np.int32)), _lib.ptr(np.zeros(4, np.int32)),
                                _lib.ptr(np.zeros((4, 2), np.int64)), _lib.ptr(cnt))
    assert rc == -1
    # band solve: window kernel outside its band range, two-way mode on a short matrix
    low = np.ones((64, 9)); rhs = np.ones(64); x = np.zeros(64)
    assert L.bh_ba_debug_band_solve(ctx.h, _lib.ptr(low), _lib.ptr(rhs), 64, 8, 1, _lib.ptr(x), None) == -1
    low = np.ones((600, 246)); rhs = np.ones(600); x = np.zeros(600)
    assert L.bh_ba_debug_band_solve(ctx.h, _lib.ptr(low), _lib.ptr(rhs), 600, 245, 3, _lib.ptr(x), None) == -1
    # all-reduce hook registration: bad rank; a sharded problem without edges
    from boslam_amd import ba, synth
    s = ba.LocalBA(ctx, np.array([synth.FX, synth.FY, synth.CX, synth.CY, 0.0]))
    cb = ba._AR_FN(lambda user, dev, count, op: 0)
    assert L.bh_ba_set_allreduce(s.h, C.cast(cb, C.c_void_p), None, 3, 2) == -1
    assert L.bh_ba_set_allreduce(s.h, C.cast(cb, C.c_void_p), None, 0, 2) == 0
    with pytest.raises(Exception):
        s.set_problem(np.eye(3)[None], np.zeros((1, 3)), np.zeros(1, np.uint8), np.zeros((0, 3)), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2)))
    assert L.bh_ba_set_allreduce(s.h, None, None, 0, 1) == 0               # back to the single-GPU behaviour
    s.close()
    # bh_ctx_copy round trip
    a = np.arange(100, dtype=np.float64); b = np.zeros(100)
    import torch
    d = torch.zeros(100, dtype=torch.float64, device="cuda")
    assert L.bh_ctx_copy(ctx.h, C.c_void_p(d.data_ptr()), _lib.ptr(a), 800, 1) == 0
    assert L.bh_ctx_copy(ctx.h, _lib.ptr(b), C.c_void_p(d.data_ptr()), 800, 0) == 0 and np.array_equal(a, b)


def test_dense_corner_frame_takes_the_rescan_path(ctx):
    """White noise: thousands of NMS survivors per level, far more non-empty score-map vectors than k_select records per wave, so the
    compaction falls back to its second full scan; results must still equal the oracle bit for bit."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (2, 480, 640), dtype=np.uint8)
    img[1, :, :320] = 128                                            # half flat, half noise
    params = frontend.make_params(nfeatures=1000)
    kps, desc, n = frontend.orb_detect_describe(ctx, img, params=params)
    P = oracle.orb_params(nfeatures=1000)
    for f in range(2):
        okp, odesc, rc = oracle.orb_detect_describe(img[f], None, P)
        assert rc == 0 and n[f] == len(okp) and np.array_equal(desc[f, :n[f]], odesc)
        for fld in ("x", "y", "angle", "response", "octave"):
            assert np.array_equal(kps[fld][f, :n[f]].view(np.uint32), okp[fld].view(np.uint32)), fld


def test_mp_add_observation_matches_the_reference_loop(ctx):
    """bh_mp_add_observation against MapPoint.add_observation (slam/nodes.py:142-156) restated in numpy: stored descriptors, counts,
    the medoid descriptor exactly; the running mean of the viewing direction to the last bit but one (numpy's norm is a BLAS dot)."""
    import oracle
    from boslam_amd import frontend
    rng = np.random.default_rng(4)
    M = 300
    base = rng.integers(0, 256, (M, 32), dtype=np.uint8)
    cnt = rng.integers(0, 11, M).astype(np.int32)
    obs = np.zeros((M, 10, 32), np.uint8)
    for i in range(M):
        for k in range(cnt[i]):
            d = base[i].copy(); flip = rng.integers(0, 256, 3); d[flip % 32] ^= (1 << (flip % 8)).astype(np.uint8)
            obs[i, k] = d
    cnt[:3] = [0, 9, 10]
    n_obs = (cnt + rng.integers(0, 5, M)).astype(np.int32)
    normal = rng.normal(0, 1, (M, 3)); normal /= np.linalg.norm(normal, axis=1)[:, None]
    feat = base.copy()
    state = (obs, cnt, n_obs, normal, feat)
    ref = [([obs[i, k].copy() for k in range(cnt[i])], int(n_obs[i]), normal[i].copy(), feat[i].copy()) for i in range(M)]
    for step in range(3):                                                       # three keyframes in a row: lists fill up to ten and stay there
        nd = base ^ rng.integers(0, 2, (M, 32), dtype=np.uint8)
        nn = rng.normal(0, 2, (M, 3))
        state = frontend.mp_add_observations(ctx, *state, nd, nn)
        for i in range(M):
            o, no, nr, ft = oracle.mp_add_observation(ref[i][0], ref[i][1], ref[i][2], ref[i][3], nd[i], nn[i])
            ref[i] = (o, no, nr, ft)
            assert state[1][i] == len(o) and state[2][i] == no
            assert all(np.array_equal(state[0][i, k], o[k]) for k in range(len(o)))
            assert np.array_equal(state[4][i], ft), (step, i)
            assert np.abs(state[3][i] - nr).max() <= 4e-16


def test_kf_add_map_points_matches_the_reference_loop(ctx):
    """bh_kf_add_map_points against the per-feature loop of CovisibilityGraph.add_kf (slam/covisibility_graph.py:123-136) and MapPoint.__init__ /
    add_observation (slam/nodes.py:105-123, 142-156) restated in numpy, three keyframes in a row: ids in feature order, stored descriptors,
    observing keyframes, counts and the medoid exactly; directions to 4e-16 (numpy's norm is a BLAS dot); features tied to the same map point
    are applied in feature order; erased map points are re-created; a feature at the camera centre is skipped."""
    import oracle
    from boslam_amd import map_points
    rng = np.random.default_rng(8)
    tab = map_points.MapPointTable(4096)
    ref = {}                                                         # id -> dict(obs, obs_kf, n_obs, normal, feat, pt3d, first_kf)
    cnt = 0
    for kf_id, N in ((10, 500), (11, 600), (12, 550)):
        t = rng.normal(0, 0.1, 3)
        des = rng.integers(0, 256, (N, 32), dtype=np.uint8)
        cloud = np.stack([rng.uniform(-2, 2, N), rng.uniform(-1.5, 1.5, N), rng.uniform(1, 5, N)], 1)
        d2m = -np.ones(N, np.int32)
        if cnt:
            tied = rng.random(N) < 0.6
            d2m[tied] = rng.integers(0, cnt, int(tied.sum()))        # with replacement: several features per map point do occur
            for i in np.nonzero(tied)[0][:40]:
                des[i] = ref[int(d2m[i])]["feat"] ^ rng.integers(0, 2, 32).astype(np.uint8) if int(d2m[i]) in ref else des[i]
        cloud[7] = t                                                 # |n| == 0: skipped (covisibility_graph.py:128)
        if kf_id == 12:
            for dead in (3, 17, 300):
                tab.erase(dead); ref.pop(dead, None)
            d2m[20] = 3; d2m[21] = 17                                 # tied to erased map points: re-created as new ones
        want_d2m = d2m.copy(); want_act = np.zeros(N, np.int32)
        for i in range(N):                                           # the reference loop
            n = t - cloud[i]
            norm = np.linalg.norm(n)
            if norm > 0.0:
                id_mp = int(d2m[i])
                if id_mp == -1 or id_mp not in ref:
                    nn = n / norm
                    ref[cnt] = dict(obs=[des[i].copy()], obs_kf=[kf_id], n_obs=1, normal=nn / np.linalg.norm(nn), feat=des[i].copy(), pt3d=cloud[i].copy(), first_kf=kf_id)
                    want_d2m[i] = cnt; want_act[i] = 1; cnt += 1
                else:
                    r = ref[id_mp]
                    had = len(r["obs"])
                    r["obs"], r["n_obs"], r["normal"], r["feat"] = oracle.mp_add_observation(r["obs"], r["n_obs"], r["normal"], r["feat"], des[i], n)
                    if len(r["obs"]) > had:
                        r["obs_kf"].append(kf_id)
                    want_act[i] = 2
        got_d2m, act = tab.add_kf(ctx, kf_id, t, des, cloud, d2m)
        assert np.array_equal(got_d2m, want_d2m) and np.array_equal(act, want_act) and tab.cnt == cnt
        for mid, r in ref.items():
            k = len(r["obs"])
            assert tab.alive[mid] == 1 and tab.obs_cnt[mid] == k and tab.n_obs[mid] == r["n_obs"] and tab.first_kf[mid] == r["first_kf"], mid
            assert all(np.array_equal(tab.obs_desc[mid, j], r["obs"][j]) for j in range(k)) and list(tab.obs_kf[mid, :k]) == r["obs_kf"], mid
            assert np.array_equal(tab.feat[mid], r["feat"]) and np.array_equal(tab.pt3d[mid], r["pt3d"]), mid
            assert np.abs(tab.normal[mid] - r["normal"]).max() <= 4e-16, mid
        kp_arr = rng.integers(0, 480, (N, 2))
        m2k = map_points.make_mp2kp(got_d2m, kp_arr)
        assert all(np.array_equal(m2k[int(m)], kp_arr[np.nonzero(got_d2m == m)[0][-1]]) for m in np.unique(got_d2m))
    assert (np.bincount(want_d2m[want_act == 2]).max() > 1)           # the duplicate case was exercised
    small = map_points.MapPointTable(8)
    with pytest.raises(Exception):
        small.add_kf(ctx, 1, np.zeros(3), rng.integers(0, 256, (20, 32), dtype=np.uint8), rng.uniform(1, 2, (20, 3)))
    assert small.alive.sum() == 8                                    # the table filled up to its capacity before the call reported it
