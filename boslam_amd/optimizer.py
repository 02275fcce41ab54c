"""Drop-in for the reference's slam/optimizer.py (same names, arguments and error behaviour), backed by the HIP
bundle-adjustment kernels.  The g2o graph objects the reference builds per call are replaced by flat arrays; the
arithmetic (EdgeProjectP2MC residual/Jacobians, Huber, Schur complement, Levenberg) runs on the GPU.

Mirrors
  CamOnlyBA(pixels, points, cam, R=None, t=None, information, robust_kernel, num_iter=10, flag_verbose=False)
      slam/optimizer.py:82-142 — returns (True, R, t, inliers) or (False, None, None, None) on any failure
  BundleAdjustment(cam).add_pose / add_point / add_edge / optimize / get_pose / get_point / outlier_edges
      slam/optimizer.py:8-79
"""
import numpy as np
from . import _lib
from .ba import LocalBA, pose_only_ba_batch, cam_to_K, CHI2_SIG_VALUE, HUBER_DELTA

chi2_sig_value = CHI2_SIG_VALUE


class RobustKernelHuber:
    """g2o.RobustKernelHuber(delta) stand-in: only carries delta."""

    def __init__(self, delta=HUBER_DELTA):
        self._delta = float(delta)

    def delta(self):
        return self._delta


class SE3Quat:
    """The subset of g2o.SE3Quat the reference uses (optimizer.py:25-26,52-53; tracking.py:101): R,t container,
    `pose * point`, orientation()/position()/matrix()."""

    def __init__(self, R=None, t=None):
        self.R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64).reshape(3, 3)
        self.t = np.zeros(3) if t is None else np.asarray(t, dtype=np.float64).reshape(-1)[:3]

    def __mul__(self, p):
        return self.R @ np.asarray(p, dtype=np.float64).reshape(3) + self.t

    def orientation(self):
        return self.R

    def position(self):
        return self.t

    def matrix(self):
        m = np.eye(4); m[:3, :3] = self.R; m[:3, 3] = self.t
        return m


class CameraParameters:
    """g2o.CameraParameters(f, principal_point, baseline).cam_map — tracking.py:36,102; optimizer.py:53."""

    def __init__(self, focal_length, principal_point, baseline):
        self.focal_length = float(focal_length)
        self.principal_point = (float(principal_point[0]), float(principal_point[1]))
        self.baseline = float(baseline)

    def cam_map(self, p):
        p = np.asarray(p, dtype=np.float64).reshape(3)
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.array([self.focal_length * p[0] / p[2] + self.principal_point[0],
                             self.focal_length * p[1] / p[2] + self.principal_point[1]])


def _delta_of(robust_kernel):
    if robust_kernel is None:
        return 0.0
    if hasattr(robust_kernel, "delta"):
        d = robust_kernel.delta
        return float(d() if callable(d) else d)
    return HUBER_DELTA


def _check_information(information):
    if information is not None and not np.array_equal(np.asarray(information, dtype=np.float64), np.identity(2)):
        raise ValueError("only the identity information matrix is supported (the only value the reference passes)")


_DEFAULT_KERNEL = RobustKernelHuber(np.sqrt(chi2_sig_value))


def CamOnlyBA(pixels, points, cam, R=None, t=None, information=np.identity(2), robust_kernel=_DEFAULT_KERNEL,
              num_iter=10, flag_verbose=False, ctx=None):
    """slam/optimizer.py:82-142.  R, t are ignored exactly as the reference ignores them (`if True:` at :90-91)."""
    try:
        _check_information(information)
        ctx = ctx or _lib.default_context()
        pixels = np.asarray(pixels, dtype=np.float64).reshape(-1, 2)        # int64 kp_arr reaches g2o as double
        points = np.asarray(points, dtype=np.float64).reshape(-1, 3)
        n = min(len(pixels), len(points))                                     # zip() semantics of optimizer.py:110
        Rb, tb, chi2, inl, st = pose_only_ba_batch(ctx, pixels[None, :n], points[None, :n], [n], cam_to_K(cam),
                                                   iters=num_iter, huber_delta=_delta_of(robust_kernel),
                                                   chi2_thr=chi2_sig_value)
        if st[0] != 0:
            raise FloatingPointError("pose-only BA diverged")
        return True, Rb[0], tb[0], np.nonzero(inl[0, :n])[0]
    except Exception as e:          # the reference swallows every exception here (optimizer.py:140-142)
        print(e)
        return False, None, None, None


class BundleAdjustment:
    """slam/optimizer.py:8-79 with the same vertex-id scheme (pose 2*id, point 2*id+1)."""

    def __init__(self, cam, ctx=None):
        self.cam = cam
        self.ctx = ctx or _lib.default_context()
        self._pose_ids, self._pose_R, self._pose_t, self._pose_fixed, self._pose_K = [], [], [], [], None
        self._pose_index = {}
        self._pt_ids, self._pts = [], []
        self._pt_index = {}
        self._edges = []                  # (pose index, point index, u, v, delta)
        self._solver = None
        self._result = None

    # -- graph construction -----------------------------------------------------------------------------------
    def add_pose(self, kf, cam, fixed=False):
        pose_id, R, t = kf.idf(), np.asarray(kf.Rf(), dtype=np.float64), np.asarray(kf.tf(), dtype=np.float64)
        if t.ndim == 2:
            t = t[:, 0]
        K = cam_to_K(cam)
        if self._pose_K is None:
            self._pose_K = K
        elif not np.array_equal(self._pose_K, K):
            raise ValueError("all poses of one BundleAdjustment must share the intrinsics")
        self._pose_index[pose_id] = len(self._pose_ids)
        self._pose_ids.append(pose_id); self._pose_R.append(R.reshape(3, 3)); self._pose_t.append(t.reshape(3))
        self._pose_fixed.append(bool(fixed))

    def add_point(self, mp, fixed=False, marginalized=True):
        if fixed or not marginalized:
            raise ValueError("the reference only adds free, marginalised points (optimizer.py:40-47)")
        point_id, point = mp.idf(), np.asarray(mp.pt3df(), dtype=np.float64).reshape(-1)[:3]
        self._pt_index[point_id] = len(self._pt_ids)
        self._pt_ids.append(point_id); self._pts.append(point)

    def add_edge(self, mp, kf, measurement, information=np.identity(2), robust_kernel=_DEFAULT_KERNEL):
        _check_information(information)
        point_id, pose_id = mp.idf(), kf.idf()
        pose = SE3Quat(kf.Rf(), kf.tf())
        pixel = self.cam.cam_map(pose * np.asarray(mp.pt3d, dtype=np.float64).reshape(-1)[:3])
        if 0 <= pixel[0] < 640 and 0 <= pixel[1] < 480:                       # optimizer.py:54
            m = np.asarray(measurement, dtype=np.float64).reshape(-1)
            self._edges.append((self._pose_index[pose_id], self._pt_index[point_id], m[0], m[1], _delta_of(robust_kernel)))

    # -- optimisation -------------------------------------------------------------------------------------------
    def optimize(self, max_iterations=20, flag_verbose=True):
        if not self._pose_ids:
            return
        deltas = {e[4] for e in self._edges}
        if len(deltas) > 1:
            raise ValueError("one robust kernel per graph (the reference shares a single default instance)")
        delta = deltas.pop() if deltas else 0.0
        self._solver = LocalBA(self.ctx, self._pose_K)
        ep = np.array([e[0] for e in self._edges], np.int32); eq = np.array([e[1] for e in self._edges], np.int32)
        meas = np.array([[e[2], e[3]] for e in self._edges], np.float64).reshape(-1, 2)
        pts = np.array(self._pts, np.float64).reshape(-1, 3)
        self._solver.set_problem(np.array(self._pose_R), np.array(self._pose_t), np.array(self._pose_fixed, np.uint8),
                                 pts, ep, eq, meas)
        self._solver.optimize(max_iterations, delta, verbose_errors=flag_verbose)
        R, t = self._solver.poses()
        self._result = dict(R=R, t=t, pts=self._solver.points(), chi2=self._solver.chi2())

    def _ensure(self):
        if self._result is None:
            pts = np.array(self._pts, np.float64).reshape(-1, 3)
            self._result = dict(R=np.array(self._pose_R), t=np.array(self._pose_t), pts=pts,
                                chi2=np.zeros(len(self._edges)))

    def get_pose(self, pose_id):
        self._ensure()
        i = self._pose_index[pose_id]
        return self._result["R"][i], self._result["t"][i]

    def get_point(self, point_id):
        self._ensure()
        return self._result["pts"][self._pt_index[point_id]].T

    def outlier_edges(self):
        self._ensure()
        bad = []
        for (pi, qi, _, _, _), c in zip(self._edges, self._result["chi2"]):
            if c > chi2_sig_value:
                bad.append((self._pose_ids[pi], self._pt_ids[qi]))
        return bad
