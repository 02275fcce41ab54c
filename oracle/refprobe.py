"""Differential against the REAL reference libraries, when a machine has them.

TEST INFRASTRUCTURE (bench.py's cpu_baseline leg and tests/test_reference_libs.py only).  The reference's arithmetic lives in
cv2 / g2o, which are absent from this image; everywhere they are absent this module reports exactly that and parity stays
"oracle only" (unpinned).  Where `import cv2` / `import g2o` work, the reference's own calls are issued with the reference's own
parameters and their results are compared with the CPU oracle's:
  cv2.ORB_create(**orb_params).detectAndCompute(frame, mask=(depth > 0) * 255)     camera.py:49-62,141
  cv2.BFMatcher_create(cv2.NORM_HAMMING, crossCheck=True).match(q, t)              slam/tracking.py:45,56
  cv2.solvePnPRansac(obj, img, K, dist, flags=cv2.SOLVEPNP_ITERATIVE)              experiments/pnp_two_way_tracking.py:57-61
  the g2o graph of CamOnlyBA                                                       slam/optimizer.py:82-142
Keypoints are compared as sets per level in the canonical order (level, y, x): OpenCV's order inside a level is whatever
std::nth_element leaves."""
import importlib
import time
import numpy as np


def probe():
    out = {}
    for name in ("cv2", "g2o"):
        try:
            m = importlib.import_module(name)
            out[name] = getattr(m, "__version__", "present")
        except Exception as e:                                   # noqa: BLE001 - ImportError, or a broken build
            out[name] = None
            out[name + "_error"] = f"{type(e).__name__}: {e}"
    return out


def canonical(kx, ky, octave):
    return np.lexsort((kx, ky, octave))


def cv2_orb(cv2, frame_bgr, depth, nfeatures):
    """The reference's detector call (camera.py:49-62,141) -> structured keypoints + descriptors in canonical order."""
    orb = cv2.ORB_create(nfeatures=int(nfeatures), scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, patchSize=31,
                         fastThreshold=20)
    mask = ((depth > 0) * 255).astype(np.uint8)
    kp, des = orb.detectAndCompute(frame_bgr, mask=mask)
    if des is None:
        return np.zeros((0, 6), np.float32), np.zeros((0, 32), np.uint8)
    a = np.array([[k.pt[0], k.pt[1], k.size, k.angle, k.response, k.octave] for k in kp], dtype=np.float32)
    o = canonical(a[:, 0], a[:, 1], a[:, 5])
    return a[o], des[o]


def diff_orb(cv_kp, cv_des, o_kps, o_des):
    """-> dict(identical, n_cv2, n_oracle, common_keypoints, identical_descriptors_among_common)."""
    ok = np.stack([o_kps["x"], o_kps["y"], o_kps["size"], o_kps["angle"], o_kps["response"], o_kps["octave"].astype(np.float32)], 1)
    oo = canonical(ok[:, 0], ok[:, 1], ok[:, 5])
    ok, od = ok[oo], o_des[oo]
    key = lambda a: {(int(r[5]), float(r[0]), float(r[1])): i for i, r in enumerate(a)}      # noqa: E731
    kc, ko = key(cv_kp), key(ok)
    common = sorted(set(kc) & set(ko))
    same_desc = sum(int(np.array_equal(cv_des[kc[k]], od[ko[k]])) for k in common)
    same_angle = sum(int(cv_kp[kc[k], 3] == ok[ko[k], 3]) for k in common)
    identical = len(cv_kp) == len(ok) == len(common) and same_desc == len(common) and same_angle == len(common)
    return dict(identical=bool(identical), n_cv2=len(cv_kp), n_oracle=len(ok), common_keypoints=len(common),
                identical_descriptors_among_common=same_desc, identical_angles_among_common=same_angle)


def cv2_match(cv2, q, t):
    m = cv2.BFMatcher_create(cv2.NORM_HAMMING, crossCheck=True).match(q, t)
    return np.array([[x.queryIdx, x.trainIdx, int(x.distance)] for x in m], dtype=np.int64).reshape(-1, 3)


def g2o_cam_only_ba(g2o, pixels, points, K5, num_iter=10, chi2_thr=7.815):
    """The graph slam/optimizer.py:82-142 builds, with g2opy's API: one free VertexCam at identity, fixed marginalised points,
    EdgeProjectP2MC, information I, Huber(sqrt(7.815)), Levenberg over CSparse; returns (R, t, inlier mask)."""
    opt = g2o.SparseOptimizer()
    opt.set_algorithm(g2o.OptimizationAlgorithmLevenberg(g2o.BlockSolverSE3(g2o.LinearSolverCSparseSE3())))
    pose = g2o.SE3Quat(np.eye(3), np.zeros(3))
    cam = g2o.SBACam(pose.orientation(), pose.position())
    cam.set_cam(float(K5[0]), float(K5[1]), float(K5[2]), float(K5[3]), float(K5[4]))
    vc = g2o.VertexCam(); vc.set_id(0); vc.set_estimate(cam); opt.add_vertex(vc)
    kern = g2o.RobustKernelHuber(np.sqrt(chi2_thr))
    for i, (px, pt) in enumerate(zip(pixels, points)):
        vp = g2o.VertexSBAPointXYZ(); vp.set_id(i + 1); vp.set_estimate(pt); vp.set_marginalized(True); vp.set_fixed(True)
        opt.add_vertex(vp)
        e = g2o.EdgeProjectP2MC(); e.set_vertex(0, opt.vertex(i + 1)); e.set_vertex(1, opt.vertex(0))
        e.set_measurement(px); e.set_information(np.identity(2)); e.set_robust_kernel(kern)
        opt.add_edge(e)
    opt.initialize_optimization(); opt.set_verbose(False); opt.optimize(int(num_iter))
    inl = np.zeros(len(pixels), bool)
    for e in opt.edges():
        if e.chi2() <= chi2_thr:
            inl[e.vertices()[0].id() - 1] = True
    m = opt.vertex(0).estimate().matrix()
    return m[:3, :3], m[:3, 3], inl


def run(bgr, depth, nfeatures, o_kps, o_desc, o_nk):
    """bench.py's probe: time + diff the reference calls on the given frames (the oracle's results for the same frames are
    passed in).  Returns a JSON-able dict; {"cv2": "absent", "g2o": "absent", ...} on this image."""
    import oracle
    have = probe()
    out = {"cv2": have["cv2"] or "absent", "g2o": have["g2o"] or "absent"}
    if not have["cv2"] and not have["g2o"]:
        out["note"] = "reference libs absent - parity vs oracle only (unpinned)"
        return out
    if have["cv2"]:
        import cv2
        n = len(bgr)
        t0 = time.perf_counter()
        res = [cv2_orb(cv2, bgr[f], depth[f], nfeatures) for f in range(n)]
        t_orb = time.perf_counter() - t0
        diffs = [diff_orb(res[f][0], res[f][1], o_kps[f, :o_nk[f]], o_desc[f, :o_nk[f]]) for f in range(n)]
        t0 = time.perf_counter()
        same_m = 0
        for f in range(1, n):
            cm = cv2_match(cv2, o_desc[f, :o_nk[f]], o_desc[f - 1, :o_nk[f - 1]])
            oq, ot, od = oracle.match_crosscheck(o_desc[f, :o_nk[f]], o_desc[f - 1, :o_nk[f - 1]])
            same_m += int(np.array_equal(cm, np.stack([oq, ot, od], 1).astype(np.int64)))
        t_m = time.perf_counter() - t0
        out["cv2_orb"] = {"frames": n, "frames_identical_to_oracle": sum(d["identical"] for d in diffs),
                          "mean_common_keypoints": float(np.mean([d["common_keypoints"] for d in diffs])),
                          "mean_n_cv2": float(np.mean([d["n_cv2"] for d in diffs])), "mean_n_oracle": float(np.mean([d["n_oracle"] for d in diffs])),
                          "frames_per_s_single_process": round(n / t_orb, 1)}
        out["cv2_match"] = {"pairs": n - 1, "pairs_identical_to_oracle": same_m, "pairs_per_s": round((n - 1) / max(t_m, 1e-9), 1)}
    if have["g2o"]:
        import g2o
        from boslam_amd import synth
        p = synth.pose_problem(n=250, seed=11)
        t0 = time.perf_counter()
        R, t, inl = g2o_cam_only_ba(g2o, p["pix"], p["pts"], p["K"])
        dt = time.perf_counter() - t0
        o = oracle.pose_only_ba(p["pix"], p["pts"], p["K"])
        out["g2o_cam_only_ba"] = {"max_abs_pose_diff_vs_oracle": float(max(np.abs(R - o["R"]).max(), np.abs(t - o["t"]).max())),
                                  "inlier_sets_equal": bool(np.array_equal(inl, o["inlier"])), "ms": round(dt * 1e3, 2)}
    return out
