"""Checker-side restatement of experiments/pnp_two_way_tracking.py (BASELINE configs[0]), written from the script itself.

TEST INFRASTRUCTURE (see oracle/__init__.py): only tests/ may import this.  It exists so that the product's replay
(boslam_amd/two_way_tracking.py) is NOT the state machine on both sides of its own test: VERDICT r03 weak 1(b) - the golden log
was generated through the product's `replay`, so a slip in the fusion lines would have cancelled.  Here the script's control
flow is kept as it stands in the reference - a `while` loop with `continue`, frame objects that carry `des / kp_arr / cloud_kp`
and a `transform2global` method (camera.py:32-43), `zip(*...)` index extraction, a Python `set` union of the two inlier lists -
and shares no line with the product.  PARITY UNPINNED against OpenCV like the rest of oracle/ (the matcher / PnP back ends are
the oracle's), but the state machine is now checked against an independent reading of the reference.
Line numbers: experiments/pnp_two_way_tracking.py unless a file is named."""
import numpy as np


def _normalize_t_shape(t):
    # utils.py:47-51 as camera.py:34 uses it: a (3, 1) translation becomes its first column, a (3,) one stays - either broadcasts over (N, 3)
    t = np.asarray(t, dtype=np.float64)
    return t[:, 0] if len(t.shape) > 1 else t


class _FrameOb:
    """camera.py:12-43, the fields the script touches."""

    def __init__(self, des, kp_arr, cloud_kp):
        self.des = des
        self.kp_arr = kp_arr
        self.cloud_kp = cloud_kp

    def transform2global(self, R, t, prev_cloud_kp=None, new_inds_for_old=None, log=None):
        t = _normalize_t_shape(t)                                                        # camera.py:34
        self.cloud_kp = np.matmul(self.cloud_kp, R) + t                                  # camera.py:35
        if log is not None:                                                              # camera.py:39-40
            log['3d_point_diff'] = np.average(np.linalg.norm(self.cloud_kp[new_inds_for_old] - prev_cloud_kp, axis=1))
        if prev_cloud_kp is not None and new_inds_for_old is not None:                   # camera.py:42-43
            self.cloud_kp[new_inds_for_old] = prev_cloud_kp


class _Match:
    __slots__ = ("queryIdx", "trainIdx", "distance")

    def __init__(self, q, t, d):
        self.queryIdx, self.trainIdx, self.distance = int(q), int(t), float(d)


def run_script(frames, match, solve_pnp_ransac, rodrigues):
    """frames: list of (des u8[N][32] or None, kp_arr i64[N][2], cloud_kp f64[N][3]) in the order camera.get() returns them.
    match(q_des, t_des) -> (queryIdx, trainIdx, distance) arrays; solve_pnp_ransac(obj, img) -> (ok, rvec, tvec, inliers);
    rodrigues(rvec) -> 3x3.  Returns one dict per frame: what the script logs (:49,79,80,88 + camera.py:40), its `state`
    verdict (1 tracked, 0 initialised, -1 too few matches :44, -2 a PnP failed :63, -3 no descriptors :38) and the fused rvec / tvec (:71-72)."""
    out = []
    prev_frame_ob = None
    state = 0                                                                            # :20
    i_frame = -1
    camera_state = np.eye(3), np.zeros((3, 1))                                           # :29
    it = iter(frames)
    while True:                                                                          # :31
        i_frame += 1
        try:
            des, kp_arr, cloud_kp = next(it)                                             # :34 camera.get()
        except StopIteration:
            break
        frame_ob = _FrameOb(des, np.array(kp_arr), np.array(cloud_kp, dtype=np.float64))
        log = {}
        rec = {'state': 0, 'log': log, 'rvec': None, 'tvec': None}
        out.append(rec)

        if state == 1:                                                                   # :37
            rec['state'] = 1
            if prev_frame_ob.des is None or frame_ob.des is None:                        # :38
                state = 0
                rec['state'] = -3
                continue
            qi, ti, dd = match(prev_frame_ob.des, frame_ob.des)                          # :42
            matches = [_Match(q, t, d) for q, t, d in zip(qi, ti, dd)]
            matches = [m for m in matches if m.distance < 20]                            # :43
            if len(matches) < 10:                                                        # :44
                state = 0
                rec['state'] = -1
                continue
            log['num_matches'] = len(matches)                                            # :49
            inds_prev, inds = zip(*((_.queryIdx, _.trainIdx) for _ in matches))          # :50
            a = np.ascontiguousarray(prev_frame_ob.cloud_kp[inds_prev, :])               # :51
            b = np.ascontiguousarray(frame_ob.kp_arr[inds, :].astype(np.float64))        # :52
            a1 = np.ascontiguousarray(prev_frame_ob.kp_arr[inds_prev, :].astype(np.float64))      # :54
            b1 = np.ascontiguousarray(frame_ob.cloud_kp[inds, :])                        # :55
            is_ok_ab, rvec_ab, tvec_ab, inliers_ab = solve_pnp_ransac(a, b)              # :57
            is_ok_ba, rvec_ba, tvec_ba, inliers_ba = solve_pnp_ransac(b1, a1)            # :60
            if not is_ok_ab or not is_ok_ba:                                             # :63
                state = 0
                rec['state'] = -2
                continue
            rvec_ab = rodrigues(rvec_ab)                                                 # :68
            rvec_ba = rodrigues(rvec_ba)                                                 # :69
            rvec = (rvec_ab + np.linalg.inv(rvec_ba)) / 2.0                              # :71
            tvec = (np.reshape(tvec_ab, (3, 1)) - np.reshape(tvec_ba, (3, 1))) / 2.0     # :72 (OpenCV returns (3, 1))
            camera_state = np.matmul(camera_state[0], rvec), camera_state[1] + tvec      # :75
            inliers_ab = np.asarray(inliers_ab); inliers_ba = np.asarray(inliers_ba)
            inliers = np.asarray(list(set(inliers_ab.flatten()).union(inliers_ba.flatten())), dtype=np.int64)      # :77
            log['num_inliers'] = len(inliers_ab)                                         # :79
            log['norm_tvec'] = np.linalg.norm(tvec)                                      # :80
            inds_prev = np.asarray(inds_prev)                                            # :82
            inds = np.asarray(inds)
            prev_cloud_kp = prev_frame_ob.cloud_kp[inds_prev[inliers], :]                # :87
            log['old_tvec_norm'] = np.average(np.linalg.norm(prev_cloud_kp, axis=1))     # :88
            new_inds_for_old = inds[inliers]                                             # :89
            frame_ob.transform2global(rvec, tvec, prev_cloud_kp, new_inds_for_old, log=log)      # :91
            rec['rvec'], rec['tvec'], rec['num_union'] = rvec, tvec[:, 0], len(inliers)

        if state == 0:                                                                   # :93
            rvec, tvec = np.eye(3), np.zeros(3)                                          # :98
            camera_state = rvec, tvec                                                    # :99
            log['norm_tvec'] = np.linalg.norm(tvec)                                      # :100
            state = 1                                                                    # :101

        prev_frame_ob = frame_ob                                                         # :105
        rec['camera_state'] = (np.array(camera_state[0]), np.array(camera_state[1]))     # :107 prints its norm
    return out
