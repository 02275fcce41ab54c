"""BASELINE configs[0]: the build's counterpart of experiments/pnp_two_way_tracking.py (the reference's own CPU-runnable case)
over the C ABI.  ORB and the consecutive-frame matches of the whole sequence run as two batched GPU calls (they do not depend
on the tracking state); the state machine of the script is then replayed frame by frame, with the two solvePnPRansac calls of
a frame batched into one launch.

Line references are to experiments/pnp_two_way_tracking.py.  Reference quirks kept on purpose:
  * camera_state[1] starts as zeros(3) (:98-99) and tvec is (3, 1), so the accumulated translation broadcasts to 3x3 (:75);
  * the current frame's keypoint cloud is moved to the "global" frame with the fused (R, t) and the matched inliers are
    overwritten with the previous frame's points (:91, camera.py:32-43) - the next frame's PnP input depends on it;
  * a failed frame leaves prev_frame_ob untouched (`continue`, :40,47,66) and the next frame re-initialises.
`backend` supplies detect/match/pnp so that the same replay runs on the GPU library and, in tests, on the CPU oracle."""
import numpy as np


def replay(desc, n, xy, cloud, match_fn, pnp2_fn, rodrigues_fn):
    """desc u8[F][S][32], n int[F], xy i64[F][S][2], cloud f64[F][S][3] (camera.py:141-142,163).
    match_fn(f_prev, f) -> (queryIdx, trainIdx, distance); pnp2_fn(a, b, b1, a1) -> ((ok, rvec, tvec, inl), (ok, rvec, tvec, inl))."""
    F = len(n)
    cloud = [np.array(cloud[f, :n[f]], dtype=np.float64) for f in range(F)]           # mutated like FrameOb.cloud_kp
    state, prev = 0, None
    cam = (np.eye(3), np.zeros((3, 1)))
    log = dict(state=np.zeros(F, np.int32), num_matches=np.zeros(F, np.int32), num_inliers=np.zeros(F, np.int32),
               num_union=np.zeros(F, np.int32), R=np.zeros((F, 3, 3)), t=np.zeros((F, 3)), cam_R=np.zeros((F, 3, 3)),
               cam_t_norm=np.zeros(F), norm_tvec=np.zeros(F), point_diff=np.zeros(F))
    for f in range(F):
        if state == 1:                                                                  # :37
            log["state"][f] = 1
            qi, ti, d = match_fn(prev, f)                                               # :42
            good = d < 20                                                               # :43
            if good.sum() < 10:                                                         # :44
                state = 0; log["state"][f] = -1
                continue
            ip, ic = qi[good], ti[good]
            log["num_matches"][f] = len(ip)
            a = np.ascontiguousarray(cloud[prev][ip]); b = np.ascontiguousarray(xy[f, ic].astype(np.float64))        # :51-52
            a1 = np.ascontiguousarray(xy[prev, ip].astype(np.float64)); b1 = np.ascontiguousarray(cloud[f][ic])      # :54-55
            (ok_ab, r_ab, t_ab, in_ab), (ok_ba, r_ba, t_ba, in_ba) = pnp2_fn(a, b, b1, a1)                           # :57-61
            if not ok_ab or not ok_ba:                                                  # :63
                state = 0; log["state"][f] = -2
                continue
            R_ab = rodrigues_fn(r_ab); R_ba = rodrigues_fn(r_ba)                        # :68-69
            R = (R_ab + np.linalg.inv(R_ba)) / 2.0                                      # :71
            t = (np.reshape(t_ab, (3, 1)) - np.reshape(t_ba, (3, 1))) / 2.0             # :72
            cam = (cam[0] @ R, cam[1] + t)                                              # :75
            inl = np.union1d(np.ravel(in_ab), np.ravel(in_ba)).astype(np.int64)         # :77
            log["num_inliers"][f] = len(np.ravel(in_ab)); log["num_union"][f] = len(inl)
            log["norm_tvec"][f] = np.linalg.norm(t)
            prev_cloud = cloud[prev][ip[inl]]                                           # :87
            new_inds = ic[inl]                                                          # :89
            cloud[f] = cloud[f] @ R + t[:, 0]                                           # :91 -> camera.py:35
            log["point_diff"][f] = np.average(np.linalg.norm(cloud[f][new_inds] - prev_cloud, axis=1))               # camera.py:40
            cloud[f][new_inds] = prev_cloud                                             # camera.py:43
            log["R"][f] = R; log["t"][f] = t[:, 0]
        if state == 0:                                                                  # :93
            cam = (np.eye(3), np.zeros(3))                                              # :98-99
            state = 1
        log["cam_R"][f] = cam[0]; log["cam_t_norm"][f] = np.linalg.norm(np.ravel(cam[1]))                            # :107
        prev = f                                                                        # :105
    return log


def run(ctx, bgr, depth, K4, nfeatures=600):
    """The GPU path.  bgr u8[F][480][640][3], depth u16[F][480][640], K4 = (fx, fy, cx, cy)."""
    from . import frontend, cv2_shim, pnp
    params = frontend.make_params(nfeatures=nfeatures)
    kps, desc, n = frontend.orb_detect_describe(ctx, bgr, depth=depth, params=params)
    xy, cloud = frontend.kp_backproject(ctx, kps, n, depth, K4[0], K4[1], K4[2], K4[3])
    F = len(n)
    qi, ti, d, nm = frontend.match_pairs(ctx, desc[:-1], n[:-1], desc[1:], n[1:])                                     # query = previous (:42)
    cached = {(f, f + 1): (qi[f, :nm[f]], ti[f, :nm[f]], d[f, :nm[f]]) for f in range(F - 1)}

    def match_fn(p, f):
        if (p, f) not in cached:                                                        # after a lost frame the pair is not consecutive
            q, t_, dd, m = frontend.match_pairs(ctx, desc[p:p + 1], n[p:p + 1], desc[f:f + 1], n[f:f + 1])
            cached[(p, f)] = (q[0, :m[0]], t_[0, :m[0]], dd[0, :m[0]])
        return cached[(p, f)]

    def pnp2_fn(a, b, b1, a1):
        S = max(len(a), 1)
        obj = np.zeros((2, S, 3)); img = np.zeros((2, S, 2))
        obj[0, :len(a)] = a; img[0, :len(b)] = b; obj[1, :len(b1)] = b1; img[1, :len(a1)] = a1
        rv, tv, inl, ni, ok = pnp.solve_pnp_ransac_batch(ctx, obj, img, np.array([len(a), len(b1)], np.int32), K4)
        return tuple((bool(ok[k]), rv[k], tv[k], inl[k][:ni[k]]) for k in range(2))

    return replay(desc, n, xy, cloud, match_fn, pnp2_fn, lambda v: cv2_shim.Rodrigues(np.asarray(v, float))[0])
