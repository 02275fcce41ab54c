"""The map-point side of CovisibilityGraph.add_kf (slam/covisibility_graph.py:119-138) over the C ABI: the map points live in one table whose
row index is the map point's id (MapPoint.cnt, slam/nodes.py:105-123), and the whole per-feature loop of a new keyframe - create the points
that are not yet in the map, add an observation (medoid descriptor, running mean of the viewing direction) to those that are - is ONE call,
bh_kf_add_map_points.  The covisibility edges (add_edge_to_cg, :136) and KeyFrame.make_mp2kp stay with the host's graph: the call returns
what they need (des2mp, the action per feature)."""
import ctypes as C
import numpy as np
from . import _lib

BH_MEM_HOST = 0
SKIPPED, CREATED, OBSERVED = 0, 1, 2


class MapPointTable:
    """Host-resident arrays of the table (tests, small maps: every call copies them to the device and back; a live system would hold
    them in HBM and pass BH_MEM_DEVICE).  cnt = the next id (MapPoint.cnt)."""

    def __init__(self, capacity):
        c = int(capacity)
        self.cap, self.cnt = c, 0
        self.alive = np.zeros(c, np.uint8)
        self.pt3d = np.zeros((c, 3)); self.normal = np.zeros((c, 3))
        self.feat = np.zeros((c, 32), np.uint8); self.obs_desc = np.zeros((c, 10, 32), np.uint8)
        self.obs_kf = np.zeros((c, 10), np.int32); self.obs_cnt = np.zeros(c, np.int32); self.n_obs = np.zeros(c, np.int32)
        self.first_kf = np.zeros(c, np.int32)

    def erase(self, id_mp):
        """CovisibilityGraph.erase_mp's effect on the table: `id_mp not in self.mps` from now on (covisibility_graph.py:128)."""
        self.alive[int(id_mp)] = 0

    def add_kf(self, ctx, kf_id, frame_t, des, cloud_kp, des2mp=None):
        """The per-feature loop of add_kf for keyframe kf_id (des u8[N][32], cloud_kp f64[N][3], des2mp int[N] or None = all -1).
        Returns (des2mp int[N] with the new ids filled in, action int[N]: SKIPPED / CREATED / OBSERVED)."""
        des = np.ascontiguousarray(des, dtype=np.uint8).reshape(-1, 32); N = len(des)
        cloud = np.ascontiguousarray(cloud_kp, dtype=np.float64).reshape(N, 3)
        d2m = -np.ones(N, np.int32) if des2mp is None else np.array(des2mp, dtype=np.int32).reshape(N)
        t = np.ascontiguousarray(np.asarray(frame_t, dtype=np.float64).reshape(3))
        action = np.zeros(N, np.int32); created = C.c_int(0)
        ctx.check(_lib.lib().bh_kf_add_map_points(ctx.h, int(kf_id), _lib.ptr(t), N, _lib.ptr(des), _lib.ptr(cloud), _lib.ptr(d2m), self.cap, self.cnt,
                                                  _lib.ptr(self.alive), _lib.ptr(self.pt3d), _lib.ptr(self.normal), _lib.ptr(self.feat), _lib.ptr(self.obs_desc),
                                                  _lib.ptr(self.obs_kf), _lib.ptr(self.obs_cnt), _lib.ptr(self.n_obs), _lib.ptr(self.first_kf), _lib.ptr(action),
                                                  C.byref(created), BH_MEM_HOST))
        self.cnt += created.value
        return d2m, action


def make_mp2kp(des2mp, kp_arr):
    """KeyFrame.make_mp2kp (slam/nodes.py:36-39): {mp id: kp_arr[i]} with later features overwriting earlier ones (the -1 key included)."""
    return {int(m): kp_arr[i] for i, m in enumerate(des2mp)}
