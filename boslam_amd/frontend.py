"""Batched front-end over the C ABI: ORB detect+describe, keypoint back-projection, Hamming cross-check matching.

Host-side mirror of what the reference does one frame at a time in RsCamera.get (camera.py:131-168) and
Tracker._track_wrt_refkf (slam/tracking.py:54-62), batched over frames.  All arithmetic happens in
libboslam_hip.so; this module only shapes arrays."""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import OrbParams, KP_DTYPE, BH_MEM_HOST, BH_MEM_DEVICE

# camera.py:49-56
REFERENCE_ORB_PARAMS = dict(nfeatures=600, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0,
                            patchSize=31, fastThreshold=20)


def make_params(**kw):
    p = dict(REFERENCE_ORB_PARAMS)
    p.update(kw)
    return OrbParams(int(p["nfeatures"]), float(p["scaleFactor"]), int(p["nlevels"]), int(p["edgeThreshold"]),
                     int(p["firstLevel"]), int(p["patchSize"]), int(p["fastThreshold"]))


def default_max_kp(params):
    return int(params.nfeatures) + 64 + int(params.nfeatures) // 8


def orb_geometry(H, W, params):
    n = params.nlevels
    w, h, pitch, off, nf = (np.zeros(n, np.int32) for _ in range(5))
    total = C.c_int(0)
    rc = _lib.lib().bh_orb_geometry(H, W, C.byref(params), _lib.ptr(w), _lib.ptr(h), _lib.ptr(pitch), _lib.ptr(off),
                                    _lib.ptr(nf), C.byref(total))
    if rc != 0:
        raise _lib.BhError(rc, "unsupported ORB parameters")
    return dict(w=w, h=h, pitch=pitch, off=off, nfeat=nf, total=total.value)


def orb_detect_describe(ctx, imgs, mask=None, depth=None, params=None, max_kp=None):
    """Host arrays in, host arrays out.  imgs u8[B][H][W][3] or u8[B][H][W]; returns (kps[B][max_kp], desc, n_kp)."""
    params = params or make_params()
    imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
    B, H, W = imgs.shape[:3]
    ch = 1 if imgs.ndim == 3 else imgs.shape[3]
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        assert mask.shape == (B, H, W)
    if depth is not None:
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        assert depth.shape == (B, H, W)
    max_kp = max_kp or default_max_kp(params)
    kps = np.zeros((B, max_kp), KP_DTYPE)
    desc = np.zeros((B, max_kp, 32), np.uint8)
    n = np.zeros(B, np.int32)
    ctx.check(_lib.lib().bh_orb_detect_describe(ctx.h, _lib.ptr(imgs), ch, _lib.ptr(mask), _lib.ptr(depth), B, H, W,
                                                C.byref(params), max_kp, _lib.ptr(kps), _lib.ptr(desc), _lib.ptr(n),
                                                BH_MEM_HOST))
    return kps, desc, n


def orb_debug_fetch(ctx, frame, H, W, params):
    g = orb_geometry(H, W, params)
    out = {k: np.zeros(g["total"], np.uint8) for k in ("pyr", "mask", "nms", "blur")}
    ctx.check(_lib.lib().bh_orb_debug_fetch(ctx.h, frame, _lib.ptr(out["pyr"]), _lib.ptr(out["mask"]),
                                            _lib.ptr(out["nms"]), _lib.ptr(out["blur"])))
    out.update(g)
    return out


def kp_backproject(ctx, kps, n_kp, depth, fx, fy, ppx, ppy):
    """camera.py:142 + camera.py:90-118 for a batch: returns (kp_arr i64[B][max_kp][2], cloud_kp f64[B][max_kp][3])."""
    kps = np.ascontiguousarray(kps)
    B, max_kp = kps.shape
    n_kp = np.ascontiguousarray(n_kp, dtype=np.int32)
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    H, W = depth.shape[1:]
    xy = np.zeros((B, max_kp, 2), np.int64)
    cloud = np.zeros((B, max_kp, 3), np.float64)
    ctx.check(_lib.lib().bh_kp_backproject(ctx.h, _lib.ptr(kps), _lib.ptr(n_kp), _lib.ptr(depth), B, H, W, max_kp,
                                           fx, fy, ppx, ppy, _lib.ptr(xy), _lib.ptr(cloud), BH_MEM_HOST))
    return xy, cloud


def match_pairs(ctx, q, nq, t, nt, out_stride=None):
    """B independent pairs.  q u8[B][Sq][32], t u8[B][St][32] -> (qidx, tidx, dist)[B][out_stride], n_match[B]."""
    q = np.ascontiguousarray(q, dtype=np.uint8); t = np.ascontiguousarray(t, dtype=np.uint8)
    B, Sq = q.shape[:2]; St = t.shape[1]
    nq = np.ascontiguousarray(nq, dtype=np.int32); nt = np.ascontiguousarray(nt, dtype=np.int32)
    out_stride = out_stride or max(1, min(Sq, St))
    qi = np.zeros((B, out_stride), np.int32); ti = np.zeros_like(qi); d = np.zeros_like(qi)
    nm = np.zeros(B, np.int32)
    ctx.check(_lib.lib().bh_match_hamming_crosscheck(ctx.h, _lib.ptr(q), _lib.ptr(nq), Sq, _lib.ptr(t), _lib.ptr(nt), St,
                                                     B, _lib.ptr(qi), _lib.ptr(ti), _lib.ptr(d), _lib.ptr(nm),
                                                     out_stride, BH_MEM_HOST))
    return qi, ti, d, nm


class DeviceFrontEnd:
    """HBM-resident batch pipeline used by bench.py: inputs and outputs are device buffers (torch tensors are used
    only as the allocator); ORB on B frames then consecutive-frame matching (query = frame i, train = frame i-1,
    the direction of slam/tracking.py:56).

    With `dist` (torch.distributed, one rank per GPU) the ranks hold consecutive shards of ONE frame sequence: rank g's frame 0
    follows rank g-1's frame B-1.  ORB needs nothing from the neighbours; the pair across the shard boundary needs the previous
    rank's last descriptor row (max_kp x 32 B) and its keypoint count, which travel point to point (RCCL send/recv over xGMI on the
    library's stream; staged through host memory for gloo rehearsals) - no collective.  Pair 0 of rank g > 0 is then matched like
    every other pair: n_match[0] / qidx[0] ... hold it (rank 0 has no pair 0)."""

    def __init__(self, ctx, B, H, W, params, max_kp=None, torch_device="cuda:0", dist=None, backend="nccl"):
        import torch
        self.torch = torch
        self.ctx, self.B, self.H, self.W, self.params = ctx, B, H, W, params
        self.max_kp = max_kp or default_max_kp(params)
        dev = torch_device
        self.kps = torch.zeros((B, self.max_kp, 6), dtype=torch.float32, device=dev)      # bh_kp = 6 x 4 bytes
        self.desc = torch.zeros((B, self.max_kp, 32), dtype=torch.uint8, device=dev)
        self.n_kp = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.qidx = torch.zeros((B, self.max_kp), dtype=torch.int32, device=dev)
        self.tidx = torch.zeros_like(self.qidx)
        self.dist = torch.zeros_like(self.qidx)
        self.n_match = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.pg, self.backend = dist, backend
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.stream = None
        if self.world > 1:
            self.halo_desc = torch.zeros((self.max_kp, 32), dtype=torch.uint8, device=dev)
            self.halo_n = torch.zeros((1,), dtype=torch.int32, device=dev)
            if backend == "nccl":
                # one stream orders the library's kernels and RCCL's send/recv: the library adopts a torch stream
                self.stream = torch.cuda.Stream(device=dev)
                ctx.set_stream(self.stream.cuda_stream)

    def _halo_exchange(self):
        """Last descriptor row + count of this rank -> rank+1; the same from rank-1 into halo_desc / halo_n."""
        torch, pg = self.torch, self.pg
        last_d, last_n = self.desc[self.B - 1], self.n_kp[self.B - 1:self.B]
        if self.backend == "nccl":
            with torch.cuda.stream(self.stream):
                ops = []
                if self.rank + 1 < self.world:
                    ops += [pg.P2POp(pg.isend, last_d, self.rank + 1), pg.P2POp(pg.isend, last_n, self.rank + 1)]
                if self.rank > 0:
                    ops += [pg.P2POp(pg.irecv, self.halo_desc, self.rank - 1), pg.P2POp(pg.irecv, self.halo_n, self.rank - 1)]
                for w in pg.batch_isend_irecv(ops):
                    w.wait()                                                   # orders the stream, does not block the host
        else:                                                                  # gloo rehearsal: host staging
            self.ctx.synchronize()
            reqs, rd, rn = [], None, None
            if self.rank + 1 < self.world:
                reqs += [pg.isend(last_d.cpu(), self.rank + 1, tag=0), pg.isend(last_n.cpu(), self.rank + 1, tag=1)]
            if self.rank > 0:
                rd = torch.empty((self.max_kp, 32), dtype=torch.uint8); rn = torch.empty((1,), dtype=torch.int32)
                reqs += [pg.irecv(rd, self.rank - 1, tag=0), pg.irecv(rn, self.rank - 1, tag=1)]
            for r in reqs:
                r.wait()
            if rd is not None:
                self.halo_desc.copy_(rd); self.halo_n.copy_(rn)
                torch.cuda.synchronize()

    def run(self, bgr, depth):
        """bgr u8[B][H][W][3], depth u16-as-int16 [B][H][W] device tensors.  Enqueues; does not synchronise."""
        L = _lib.lib()
        c = self.ctx
        ch = 1 if bgr.dim() == 3 else bgr.shape[3]
        c.check(L.bh_orb_detect_describe(c.h, _lib.ptr(bgr), ch, None, _lib.ptr(depth), self.B, self.H, self.W,
                                         C.byref(self.params), self.max_kp, _lib.ptr(self.kps), _lib.ptr(self.desc),
                                         _lib.ptr(self.n_kp), BH_MEM_DEVICE))
        if self.B > 1:
            row = self.max_kp * 32
            c.check(L.bh_match_hamming_crosscheck(
                c.h, self.desc.data_ptr() + row, self.n_kp.data_ptr() + 4, self.max_kp,
                self.desc.data_ptr(), self.n_kp.data_ptr(), self.max_kp, self.B - 1,
                self.qidx.data_ptr() + 4 * self.max_kp, self.tidx.data_ptr() + 4 * self.max_kp,
                self.dist.data_ptr() + 4 * self.max_kp, self.n_match.data_ptr() + 4, self.max_kp, BH_MEM_DEVICE))
        if self.world > 1:
            self._halo_exchange()
            if self.rank > 0:                                                  # the pair across the shard boundary
                c.check(L.bh_match_hamming_crosscheck(
                    c.h, self.desc.data_ptr(), self.n_kp.data_ptr(), self.max_kp,
                    self.halo_desc.data_ptr(), self.halo_n.data_ptr(), self.max_kp, 1,
                    self.qidx.data_ptr(), self.tidx.data_ptr(), self.dist.data_ptr(), self.n_match.data_ptr(),
                    self.max_kp, BH_MEM_DEVICE))


class DeviceTracker:
    """RsCamera.get + Tracker._track_wrt_refkf for a batch of HBM-resident frames with no host hop (camera.py:131-168, slam/tracking.py:54-89):
    ORB on B frames -> int keypoints + depth back-projection (kp_arr, cloud_kp) -> for every frame i >= 1 against frame i-1 as the
    reference keyframe: cross-check match, `distance < d_hamming_max`, >= min_matches, CamOnlyBA, >= 15 inliers (bh_track_refkf)."""

    def __init__(self, ctx, B, H, W, params, K, max_kp=None, torch_device="cuda:0"):
        import torch
        self.torch = torch
        self.ctx, self.B, self.H, self.W, self.params = ctx, B, H, W, params
        self.K = np.ascontiguousarray(K, dtype=np.float64)
        self.max_kp = max_kp or default_max_kp(params)
        dev, m = torch_device, self.max_kp
        self.kps = torch.zeros((B, m, 6), dtype=torch.float32, device=dev)
        self.desc = torch.zeros((B, m, 32), dtype=torch.uint8, device=dev)
        self.n_kp = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.kp_xy = torch.zeros((B, m, 2), dtype=torch.int64, device=dev)
        self.cloud = torch.zeros((B, m, 3), dtype=torch.float64, device=dev)
        self.qidx = torch.zeros((B, m), dtype=torch.int32, device=dev); self.tidx = torch.zeros_like(self.qidx); self.dist = torch.zeros_like(self.qidx)
        self.n_match = torch.zeros((B,), dtype=torch.int32, device=dev); self.n_inlier = torch.zeros_like(self.n_match); self.status = torch.zeros_like(self.n_match)
        self.R = torch.zeros((B, 9), dtype=torch.float64, device=dev); self.t = torch.zeros((B, 3), dtype=torch.float64, device=dev)
        self.inlier = torch.zeros((B, m), dtype=torch.uint8, device=dev)
        self.lm = None

    def set_local_map(self, mp_pts, mp_normals, mp_feat, M):
        """Device tensors of the local map's candidates per frame, in the loop order of slam/tracking.py:97-99: mp_pts / mp_normals f64[B][Sm][3],
        mp_feat u8[B][Sm][32], M int32[B].  From then on run() chains Tracker._track_local_map (bh_track_local_map) behind the reference-keyframe
        step of every frame i >= 1, on the pose that step produced - the whole of Tracker.update's two tracking stages without a host hop.  As in
        Tracker.update (`if is_ok:`, slam/tracking.py:205-212) a frame the first stage lost is not run: its local-map status is TRACK_SKIPPED, so
        "tracked" = (lm status == TRACK_OK) needs no AND with the first stage's status."""
        torch, B, m, dev = self.torch, self.B, self.max_kp, mp_pts.device
        Sm = mp_pts.shape[1]
        o = dict(gate_idx=torch.zeros((B, Sm), dtype=torch.int32, device=dev), n_gate=torch.zeros((B,), dtype=torch.int32, device=dev))
        for k in ("qidx", "tidx", "cand", "dist"):
            o[k] = torch.zeros((B, m), dtype=torch.int32, device=dev)
        for k in ("n_match", "n_inlier", "status"):
            o[k] = torch.zeros((B,), dtype=torch.int32, device=dev)
        o["R"] = torch.zeros((B, 9), dtype=torch.float64, device=dev); o["t"] = torch.zeros((B, 3), dtype=torch.float64, device=dev)
        o["inlier"] = torch.zeros((B, m), dtype=torch.uint8, device=dev)
        self.lm = dict(pts=mp_pts.contiguous(), nrm=mp_normals.contiguous(), feat=mp_feat.contiguous(), M=M.contiguous(), Sm=Sm, out=o)
        # the tensors above were produced on torch's stream; run() reads them on the library's own (non-blocking) stream: order the two once, here
        torch.cuda.current_stream(dev).synchronize()

    def run(self, bgr, depth):
        """Enqueues ORB, the back-projection and the B - 1 tracking problems; does not synchronise.  Pair i (frame i vs i-1) at row i."""
        from . import tracking
        L, c, m = _lib.lib(), self.ctx, self.max_kp
        ch = 1 if bgr.dim() == 3 else bgr.shape[3]
        c.check(L.bh_orb_detect_describe(c.h, _lib.ptr(bgr), ch, None, _lib.ptr(depth), self.B, self.H, self.W, C.byref(self.params), m,
                                         _lib.ptr(self.kps), _lib.ptr(self.desc), _lib.ptr(self.n_kp), BH_MEM_DEVICE))
        c.check(L.bh_kp_backproject(c.h, _lib.ptr(self.kps), _lib.ptr(self.n_kp), _lib.ptr(depth), self.B, self.H, self.W, m,
                                    float(self.K[0]), float(self.K[1]), float(self.K[2]), float(self.K[3]), _lib.ptr(self.kp_xy), _lib.ptr(self.cloud), BH_MEM_DEVICE))
        if self.B > 1:
            out = dict(qidx=self.qidx.data_ptr() + 4 * m, tidx=self.tidx.data_ptr() + 4 * m, dist=self.dist.data_ptr() + 4 * m, n_match=self.n_match.data_ptr() + 4,
                       R=self.R.data_ptr() + 72, t=self.t.data_ptr() + 24, inlier=self.inlier.data_ptr() + m, n_inlier=self.n_inlier.data_ptr() + 4,
                       status=self.status.data_ptr() + 4)
            tracking.track_refkf_device(c, self.desc.data_ptr() + 32 * m, self.n_kp.data_ptr() + 4, self.kp_xy.data_ptr() + 16 * m, m,
                                        self.desc.data_ptr(), self.n_kp.data_ptr(), self.cloud.data_ptr(), m, self.B - 1, self.K, out, m)
            if self.lm is not None:                              # _track_local_map on the pose _track_wrt_refkf has just set (frame.setPose, tracking.py:87)
                lm, Sm = self.lm, self.lm["Sm"]
                o = {k: v.data_ptr() + v.element_size() * (v.numel() // self.B) for k, v in lm["out"].items()}      # row 1 onwards
                tracking.track_local_map_device(c, self.desc.data_ptr() + 32 * m, self.n_kp.data_ptr() + 4, self.kp_xy.data_ptr() + 16 * m, m,
                                                self.R.data_ptr() + 72, self.t.data_ptr() + 24, None, lm["pts"].data_ptr() + 24 * Sm, lm["nrm"].data_ptr() + 24 * Sm,
                                                lm["feat"].data_ptr() + 32 * Sm, lm["M"].data_ptr() + 4, Sm, self.B - 1, self.K, o, m,
                                                stage1_status=self.status.data_ptr() + 4, width=self.W, height=self.H)


def mp_add_observations(ctx, obs_desc, obs_cnt, n_obs, normal, feat, new_desc, new_n):
    """MapPoint.add_observation (slam/nodes.py:142-156) for M map points at once; returns the updated (obs_desc, obs_cnt, n_obs, normal, feat)."""
    obs_desc = np.array(obs_desc, dtype=np.uint8).reshape(-1, 10, 32); M = len(obs_desc)
    obs_cnt = np.array(obs_cnt, dtype=np.int32); n_obs = np.array(n_obs, dtype=np.int32)
    normal = np.array(normal, dtype=np.float64).reshape(M, 3); feat = np.array(feat, dtype=np.uint8).reshape(M, 32)
    new_desc = np.ascontiguousarray(new_desc, dtype=np.uint8).reshape(M, 32); new_n = np.ascontiguousarray(new_n, dtype=np.float64).reshape(M, 3)
    ctx.check(_lib.lib().bh_mp_add_observation(ctx.h, M, _lib.ptr(obs_desc), _lib.ptr(obs_cnt), _lib.ptr(n_obs), _lib.ptr(normal), _lib.ptr(feat),
                                               _lib.ptr(new_desc), _lib.ptr(new_n), BH_MEM_HOST))
    return obs_desc, obs_cnt, n_obs, normal, feat


class LiveTracker:
    """The reference's live loop, one frame per call (camera.py:131-168 -> slam/tracking.py:54-89): the frame goes up once (pinned staging),
    ORB + back-projection + the tracking step against the PREVIOUS frame's descriptors / cloud (kept on the device, two slots) run on
    the context's stream, and only the verdict comes back: (status, R, t, n_match, n_inlier).  Nothing else crosses PCIe."""

    def __init__(self, ctx, H, W, params, K, max_kp=None, torch_device="cuda:0", use_graph=False):
        """use_graph: from the third frame on a step replays a captured HIP graph (one per slot parity: staging copies, the ~15 launches of ORB,
        back-projection and the tracking step, the verdict copy) instead of enqueuing them one by one - the per-frame path is launch-bound."""
        import torch
        self.torch = torch
        self.use_graph, self.graphs, self.steps = bool(use_graph), [None, None], 0
        self.ctx, self.H, self.W, self.params = ctx, H, W, params
        self.K = np.ascontiguousarray(K, dtype=np.float64)
        self.max_kp = m = max_kp or default_max_kp(params)
        dev = torch_device
        self.h_bgr = torch.zeros((1, H, W, 3), dtype=torch.uint8).pin_memory(); self.h_depth = torch.zeros((1, H, W), dtype=torch.uint16).pin_memory()
        self.d_bgr = torch.zeros((1, H, W, 3), dtype=torch.uint8, device=dev); self.d_depth = torch.zeros((1, H, W), dtype=torch.uint16, device=dev)
        self.kps = torch.zeros((2, m, 6), dtype=torch.float32, device=dev); self.desc = torch.zeros((2, m, 32), dtype=torch.uint8, device=dev)
        self.n_kp = torch.zeros((2,), dtype=torch.int32, device=dev)
        self.kp_xy = torch.zeros((2, m, 2), dtype=torch.int64, device=dev); self.cloud = torch.zeros((2, m, 3), dtype=torch.float64, device=dev)
        self.qidx = torch.zeros((m,), dtype=torch.int32, device=dev); self.tidx = torch.zeros_like(self.qidx); self.dist = torch.zeros_like(self.qidx)
        self.inlier = torch.zeros((m,), dtype=torch.uint8, device=dev)
        # verdict block: [status, n_match, n_inlier, pad] as int32 + R (9) + t (3) as float64, fetched with one copy
        self.verdict = torch.zeros((14,), dtype=torch.float64, device=dev); self.h_verdict = torch.zeros((14,), dtype=torch.float64).pin_memory()
        self.slot, self.have_prev = 0, False
        self.stream = torch.cuda.Stream(device=dev)             # ONE stream orders the staging copies and the library's kernels: the library adopts it
        ctx.set_stream(self.stream.cuda_stream)

    def close(self):
        if self.stream is not None:
            self.ctx.synchronize(); self.ctx.set_stream(None); self.stream = None

    def _enqueue(self, k, p, with_prev):
        """One frame's work on self.stream: staging copies, ORB into slot k, back-projection, the tracking step against slot p, the verdict copy."""
        from . import tracking
        torch, L, c, m = self.torch, _lib.lib(), self.ctx, self.max_kp
        with torch.cuda.stream(self.stream):
            self.d_bgr.copy_(self.h_bgr, non_blocking=True); self.d_depth.copy_(self.h_depth, non_blocking=True)
        c.check(L.bh_orb_detect_describe(c.h, _lib.ptr(self.d_bgr), 3, None, _lib.ptr(self.d_depth), 1, self.H, self.W, C.byref(self.params), m,
                                         C.c_void_p(self.kps.data_ptr() + 24 * m * k), C.c_void_p(self.desc.data_ptr() + 32 * m * k),
                                         C.c_void_p(self.n_kp.data_ptr() + 4 * k), BH_MEM_DEVICE))
        c.check(L.bh_kp_backproject(c.h, C.c_void_p(self.kps.data_ptr() + 24 * m * k), C.c_void_p(self.n_kp.data_ptr() + 4 * k), _lib.ptr(self.d_depth), 1,
                                    self.H, self.W, m, float(self.K[0]), float(self.K[1]), float(self.K[2]), float(self.K[3]),
                                    C.c_void_p(self.kp_xy.data_ptr() + 16 * m * k), C.c_void_p(self.cloud.data_ptr() + 24 * m * k), BH_MEM_DEVICE))
        if with_prev:
            v = self.verdict.data_ptr()
            out = dict(qidx=self.qidx.data_ptr(), tidx=self.tidx.data_ptr(), dist=self.dist.data_ptr(), n_match=v + 4, R=v + 16, t=v + 88,
                       inlier=self.inlier.data_ptr(), n_inlier=v + 8, status=v)
            tracking.track_refkf_device(c, self.desc.data_ptr() + 32 * m * k, self.n_kp.data_ptr() + 4 * k, self.kp_xy.data_ptr() + 16 * m * k, m,
                                        self.desc.data_ptr() + 32 * m * p, self.n_kp.data_ptr() + 4 * p, self.cloud.data_ptr() + 24 * m * p, m, 1, self.K, out, m)
            with torch.cuda.stream(self.stream):
                self.h_verdict.copy_(self.verdict, non_blocking=True)

    def step(self, bgr, depth):
        """bgr u8[H][W][3], depth u16[H][W] numpy arrays -> dict(status, R, t, n_match, n_inlier) or None for the first frame."""
        torch, c = self.torch, self.ctx
        self.h_bgr[0].numpy()[...] = bgr; self.h_depth[0].numpy()[...] = depth
        k, p = self.slot, self.slot ^ 1
        res = None
        if self.use_graph and self.have_prev and self.steps >= 2:
            # every buffer, launch shape and argument of a frame with this slot parity is the same as two frames ago (the library's workspaces have
            # reached their sizes: nothing allocates, nothing synchronises in the BH_MEM_DEVICE entry points) - capture once, then replay
            if self.graphs[k] is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                    self._enqueue(k, p, True)
                self.graphs[k] = g
            with torch.cuda.stream(self.stream):
                self.graphs[k].replay()
        else:
            self._enqueue(k, p, self.have_prev)
        self.steps += 1
        # every call ends synchronised - also the first one, which has no verdict to wait for: the next step() overwrites the single
        # pinned staging pair (h_bgr / h_depth), and this frame's host-to-device copy must have left it by then
        c.synchronize()
        if self.have_prev:
            hv = self.h_verdict.numpy()
            iv = hv[:2].view(np.int32)
            res = dict(status=int(iv[0]), n_match=int(iv[1]), n_inlier=int(iv[2]), R=hv[2:11].reshape(3, 3).copy(), t=hv[11:14].copy())
        self.slot, self.have_prev = p, True
        return res


def medoid_descriptors(ctx, desc, k):
    """MapPoint.add_observation's medoid choice (slam/nodes.py:142-154) for a batch: desc u8[M][10][32], k int[M] -> idx int[M]."""
    desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 10, 32)
    k = np.ascontiguousarray(k, dtype=np.int32)
    out = np.zeros(len(k), np.int32)
    ctx.check(_lib.lib().bh_medoid_descriptor(ctx.h, _lib.ptr(desc), _lib.ptr(k), len(k), _lib.ptr(out), BH_MEM_HOST))
    return out
