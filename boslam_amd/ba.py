"""Array-level bundle-adjustment entry points over the C ABI (bh_pose_only_ba, bh_ba_*).

Conventions follow the reference's g2o usage (slam/optimizer.py): poses are SBACam estimates, i.e. camera-in-world
(R, t); the projection of world point X is K * R'(X - t)."""
import ctypes as C
import os
import numpy as np
from . import _lib
from ._lib import BH_MEM_HOST

CHI2_SIG_VALUE = 7.815                      # config.py:3
HUBER_DELTA = float(np.sqrt(CHI2_SIG_VALUE))  # slam/optimizer.py:49,82


def cam_to_K(cam):
    """The two camera flavours slam/optimizer.py:27-32 accepts."""
    if hasattr(cam, "fx"):
        return np.array([cam.fx, cam.fy, cam.cx, cam.cy, cam.baseline], dtype=np.float64)
    cx, cy = cam.principal_point
    return np.array([cam.focal_length, cam.focal_length, cx, cy, cam.baseline], dtype=np.float64)


def pose_only_ba_batch(ctx, pix, pts, n, K, iters=10, huber_delta=HUBER_DELTA, chi2_thr=CHI2_SIG_VALUE):
    """B independent CamOnlyBA problems.  pix f64[B][S][2], pts f64[B][S][3], n int[B].
    Returns R[B][3][3], t[B][3], chi2[B][S], inlier bool[B][S], status int[B]."""
    pix = np.ascontiguousarray(pix, dtype=np.float64); pts = np.ascontiguousarray(pts, dtype=np.float64)
    B, S = pix.shape[:2]
    n = np.ascontiguousarray(n, dtype=np.int32)
    K = np.ascontiguousarray(K, dtype=np.float64)
    R = np.zeros((B, 3, 3)); t = np.zeros((B, 3)); chi2 = np.zeros((B, S)); inl = np.zeros((B, S), np.uint8)
    st = np.zeros(B, np.int32)
    ctx.check(_lib.lib().bh_pose_only_ba(ctx.h, _lib.ptr(pix), _lib.ptr(pts), _lib.ptr(n), B, S, _lib.ptr(K), int(iters),
                                         float(huber_delta), float(chi2_thr), _lib.ptr(R), _lib.ptr(t), _lib.ptr(chi2),
                                         _lib.ptr(inl), _lib.ptr(st), BH_MEM_HOST))
    return R, t, chi2, inl.astype(bool), st


class LocalBA:
    """One bh_ba handle: set_problem -> optimize -> get_*."""

    def __init__(self, ctx, K):
        self.ctx = ctx
        K = np.ascontiguousarray(K, dtype=np.float64)
        h = C.c_void_p()
        ctx.check(_lib.lib().bh_ba_create(ctx.h, _lib.ptr(K), C.byref(h)))
        self.h = h
        self.C = self.P = self.E = 0

    def set_problem(self, R, t, fixed, points, edge_pose, edge_point, meas):
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(-1, 9); t = np.ascontiguousarray(t, dtype=np.float64).reshape(-1, 3)
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        edge_pose = np.ascontiguousarray(edge_pose, dtype=np.int32); edge_point = np.ascontiguousarray(edge_point, dtype=np.int32)
        meas = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 2)
        self.C, self.P, self.E = len(R), len(points), len(edge_pose)
        self._points_in = points.copy()
        self.ctx.check(_lib.lib().bh_ba_set_problem(self.h, self.C, _lib.ptr(R), _lib.ptr(t), _lib.ptr(fixed), self.P,
                                                    _lib.ptr(points), self.E, _lib.ptr(edge_pose), _lib.ptr(edge_point),
                                                    _lib.ptr(meas), BH_MEM_HOST))

    def optimize(self, iters=20, huber_delta=HUBER_DELTA, verbose_errors=True):
        done = C.c_int(0)
        self.ctx.check(_lib.lib().bh_ba_optimize(self.h, int(iters), float(huber_delta), 1 if verbose_errors else 0,
                                                 C.byref(done)))
        return done.value

    def poses(self):
        R = np.zeros((self.C, 3, 3)); t = np.zeros((self.C, 3))
        self.ctx.check(_lib.lib().bh_ba_get_poses(self.h, _lib.ptr(R), _lib.ptr(t), BH_MEM_HOST))
        return R, t

    def points(self):
        p = self._points_in.copy()
        self.ctx.check(_lib.lib().bh_ba_get_points(self.h, _lib.ptr(p), BH_MEM_HOST))
        return p

    def chi2(self):
        c = np.zeros(max(self.E, 1))
        self.ctx.check(_lib.lib().bh_ba_get_chi2(self.h, _lib.ptr(c), BH_MEM_HOST))
        return c[:self.E]

    def log(self, max_rows=64):
        lg = np.zeros((max_rows, 4))
        self.ctx.check(_lib.lib().bh_ba_get_log(self.h, _lib.ptr(lg), max_rows))
        return lg

    def solver(self):
        """Name of the reduced-camera-system solver set_problem selected (bh_ba_solver)."""
        return {0: "panel", 1: "dense", 2: "window", 3: "twoway", 4: "bcr", 5: "bcr-segments"}.get(_lib.lib().bh_ba_solver(self.h), "none")

    def structure(self):
        """The structural arrays set_problem built (test hook bh_ba_debug_structure): dict name -> int32 array."""
        names = ("e_pt", "p_start", "p_edges", "c_start", "c_edges", "pairs", "j_start", "join", "cslot")
        out = {}
        cap = 1 << 16
        for w, name in enumerate(names):
            while True:
                buf = np.zeros(cap, np.int32); n = C.c_longlong(0)
                rc = _lib.lib().bh_ba_debug_structure(self.h, w, _lib.ptr(buf), cap, C.byref(n))
                if rc == -3 and cap < (1 << 28):                               # BH_ERR_CAPACITY: grow and retry
                    cap *= 4
                    continue
                self.ctx.check(rc)
                out[name] = buf[:n.value].copy()
                break
        return out

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().bh_ba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Problem(C.Structure):                          # bh_ba_problem
    _fields_ = [("C", C.c_int), ("R", C.c_void_p), ("t", C.c_void_p), ("fixed", C.c_void_p), ("P", C.c_int), ("points", C.c_void_p),
                ("E", C.c_int), ("edge_pose", C.c_void_p), ("edge_point", C.c_void_p), ("meas", C.c_void_p)]


def set_problems(ctx, windows, problems):
    """bh_ba_batch_set_problems: LocalBA.set_problem for many windows in one call - the local windows among them are built by one launch
    (a workgroup each) and one copy in either direction.  problems: sequence of (R, t, fixed, points, edge_pose, edge_point, meas)."""
    windows = list(windows)
    assert len(windows) == len(problems)
    arr = (_Problem * len(windows))()
    keep = []
    for w, (win, pr) in enumerate(zip(windows, problems)):
        R, t, fixed, points, edge_pose, edge_point, meas = pr
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(-1, 9); t = np.ascontiguousarray(t, dtype=np.float64).reshape(-1, 3)
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        edge_pose = np.ascontiguousarray(edge_pose, dtype=np.int32); edge_point = np.ascontiguousarray(edge_point, dtype=np.int32)
        meas = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 2)
        keep.append((R, t, fixed, points, edge_pose, edge_point, meas))
        win.C, win.P, win.E = len(R), len(points), len(edge_pose)
        win._points_in = points.copy()
        arr[w] = _Problem(win.C, R.ctypes.data, t.ctypes.data, fixed.ctypes.data, win.P, points.ctypes.data, win.E,
                          edge_pose.ctypes.data, edge_point.ctypes.data, meas.ctypes.data)
    hs = (C.c_void_p * len(windows))(*[w.h for w in windows])
    ctx.check(_lib.lib().bh_ba_batch_set_problems(ctx.h, len(windows), hs, arr))


class LocalBABatch:
    """bh_ba_batch: W LocalBA windows (each with a problem set) optimised together, every kernel of a Levenberg trial launched once
    for all windows (slam/local_mapping.py:46-47 runs one window at a time).  Results are read from the windows as usual."""

    def __init__(self, ctx, windows):
        self.ctx = ctx
        self.windows = list(windows)
        arr = (C.c_void_p * len(self.windows))(*[w.h for w in self.windows])
        h = C.c_void_p()
        ctx.check(_lib.lib().bh_ba_batch_create(ctx.h, len(self.windows), arr, C.byref(h)))
        self.h = h

    def optimize(self, iters=20, huber_delta=HUBER_DELTA, verbose_errors=True):
        done = np.zeros(len(self.windows), np.int32)
        self.ctx.check(_lib.lib().bh_ba_batch_optimize(self.h, int(iters), float(huber_delta), 1 if verbose_errors else 0, _lib.ptr(done)))
        return done

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().bh_ba_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------- multi-GPU global BA
_AR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int)


class _DeviceF64:
    """A raw device address presented through __cuda_array_interface__ so torch can wrap it without a copy."""

    def __init__(self, addr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(addr), False), "version": 2}


def host_exchange(t, op, dist):
    """The hook's four exchanges on a HOST tensor (the gloo rehearsal path; also what tests/test_host_cpu.py drives with two CPU ranks).
    op 0 / 1: all-reduce sum / max; op 2: reduce-scatter sum in place - `world` equal chunks, afterwards chunk `rank` holds the sum (gloo
    has no reduce-scatter: one reduce per owner); op 3: all-gather in place - chunk `rank` is this rank's contribution."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if op < 2:
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
        return t
    ch = t.numel() // world
    if op == 2:
        for dst in range(world):
            dist.reduce(t[dst * ch:(dst + 1) * ch], dst=dst, op=dist.ReduceOp.SUM)
    else:
        parts = [torch.empty(ch, dtype=t.dtype) for _ in range(world)]
        dist.all_gather(parts, t[rank * ch:(rank + 1) * ch].clone())
        for r_, p_ in enumerate(parts):
            t[r_ * ch:(r_ + 1) * ch] = p_
    return t


def segment_blocks(n_unknowns, world):
    """The keyframe segments of the sharded reduced-camera solve: blocks of 256 rows per rank, or None when a rank would hold fewer
    than two (then S is all-reduced and the solve replicated) - the decision bh_ba_set_problem takes (ba.hip: dbcr_setup)."""
    N = (int(n_unknowns) + 255) // 256
    seg = (N + world - 1) // world
    if world < 2 or seg < 2 or N - (world - 1) * seg < 2:
        return None
    return [(r * seg, min(N, (r + 1) * seg)) for r in range(world)]


DBCR_CH = 4 * 65536 + 512            # doubles a rank contributes to the gathered end blocks (ba.hip: DBCR_CH)


def exchange_bytes_per_trial(n_unknowns, bw, world, segments=True):
    """Bytes a rank RECEIVES per Levenberg trial for the reduced camera system, by kind - the arithmetic of ba.hip's own accounting
    (ba_allreduce: an all-reduce counts its buffer, a reduce-scatter the rank's chunk, an all-gather the other ranks' chunks):
      segments=True   the keyframe-segment scheme (dbcr_solve): S reduce-scattered over 256 seg world padded rows, b~ all-reduced, the
                      segments' end blocks all-gathered (DBCR_CH doubles per rank), the step's rows all-gathered;
      segments=False  the round-2 scheme: S and b~ all-reduced, the solve replicated.
    Returns dict(reduce_scatter, all_gather, allreduce, total) in bytes, or None when segments are asked for but a rank would hold fewer
    than two blocks (segment_blocks)."""
    n, w = int(n_unknowns), int(world)
    if not segments or w < 2:
        ar = 8 * n * (bw + 1) + 8 * n
        return dict(reduce_scatter=0, all_gather=0, allreduce=ar, total=ar)
    if segment_blocks(n, w) is None:
        return None
    N = (n + 255) // 256; seg = (N + w - 1) // w
    rows_pad = 256 * seg * w
    rs = 8 * rows_pad * (bw + 1) // w
    ag = 8 * DBCR_CH * (w - 1) + 8 * 256 * seg * (w - 1)
    ar = 8 * n
    return dict(reduce_scatter=rs, all_gather=ag, allreduce=ar, total=rs + ag + ar)


def make_allreduce_hook(ctx, dist, device=None, stream=None):
    """bh_allreduce_fn over torch.distributed.  With a device-capable backend ("nccl" = RCCL over xGMI) the library's device
    buffer is reduced in place; with "gloo" it is staged through host memory (CPU rehearsals).
    stream: a torch stream the context has adopted (Context.set_stream).  The collectives are then issued with it as torch's current stream -
    RCCL's work is ordered against the current stream on both sides - so they queue behind the library's kernels and the kernels after them
    queue behind the collectives: no host synchronisation (bh_ba_set_allreduce_ordered)."""
    import torch
    on_device = dist.get_backend() == "nccl"
    L = _lib.lib()

    def hook(user, dev, count, op):
        # op 0 / 1: all-reduce sum / max; op 2: reduce-scatter sum (world equal chunks, chunk `rank` receives the sum); op 3: all-gather
        # (chunk `rank` is this rank's contribution) - all in place on the library's device buffer
        try:
            count = int(count)
            world, rank = dist.get_world_size(), dist.get_rank()
            if on_device:
                import contextlib
                with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                    t = torch.as_tensor(_DeviceF64(dev, count), device=device or "cuda")
                    if op < 2:
                        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
                    else:
                        ch = count // world
                        mine = t[rank * ch:(rank + 1) * ch]
                        if op == 2:
                            dist.reduce_scatter_tensor(mine, t)             # RCCL: in place (output = the rank's chunk of the input)
                        else:
                            dist.all_gather_into_tensor(t, mine)            # RCCL: in place
                    if stream is None:
                        torch.cuda.current_stream(t.device).synchronize()
            else:                                                            # gloo rehearsal: staged through host memory
                h = np.empty(count, np.float64)
                if L.bh_ctx_copy(ctx.h, _lib.ptr(h), C.c_void_p(dev), 8 * count, 0) != 0:
                    return 2
                host_exchange(torch.from_numpy(h), op, dist)
                if L.bh_ctx_copy(ctx.h, C.c_void_p(dev), _lib.ptr(h), 8 * count, 1) != 0:
                    return 2
            return 0
        except Exception:                       # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1

    return _AR_FN(hook)


class GlobalBA(LocalBA):
    """Sharded global BA (BASELINE configs[4]): every rank holds all keyframes and the landmarks j with
    j % world == rank (plus their observations).  Per Levenberg trial: Hpp / bp / b~ and three scalars are all-reduced; the reduced
    camera system is cut by keyframe into one segment per rank - reduce-scatter of S, local elimination of the segment's interior,
    all-gather of the segments' end blocks, a small replicated chain, local back-substitution, all-gather of the step (see
    include/boslam_hip.h, bh_ba_set_allreduce).  Short or dense systems keep the round-2 scheme (S all-reduced, solve replicated)."""

    def __init__(self, ctx, K, dist, device=None):
        super().__init__(ctx, K)
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        # RCCL: ONE stream orders the library's kernels and the collectives (the context adopts a torch stream, as frontend.DeviceFrontEnd does for
        # its halo transfers): the hook enqueues, nobody synchronises.  BH_BA_HOOK_SYNC=1 keeps the synchronising hook (A/B); gloo always stages.
        self._stream = None
        if dist.get_backend() == "nccl" and os.environ.get("BH_BA_HOOK_SYNC") != "1":
            import torch
            self._stream = torch.cuda.Stream(device=device or "cuda")
            ctx.set_stream(self._stream.cuda_stream)
        self._hook = make_allreduce_hook(ctx, dist, device, stream=self._stream)          # keep the CFUNCTYPE object alive
        ctx.check(_lib.lib().bh_ba_set_allreduce(self.h, C.cast(self._hook, C.c_void_p), None, self.rank, self.world))
        if self._stream is not None:
            ctx.check(_lib.lib().bh_ba_set_allreduce_ordered(self.h, 1))

    def close(self):
        if getattr(self, "h", None) and getattr(self, "_stream", None) is not None:
            self.ctx.synchronize(); self.ctx.set_stream(None); self._stream = None
        super().close()

    def set_problem(self, R, t, fixed, points, edge_pose, edge_point, meas):
        """Takes the WHOLE problem on every rank and keeps this rank's landmark shard."""
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        edge_pose = np.ascontiguousarray(edge_pose, dtype=np.int32); edge_point = np.ascontiguousarray(edge_point, dtype=np.int32)
        meas = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 2)
        self._P_all = len(points)
        self.local_points = np.arange(self.rank, len(points), self.world)
        self.local_edges = np.nonzero(edge_point % self.world == self.rank)[0]
        super().set_problem(R, t, fixed, points[self.local_points], edge_pose[self.local_edges],
                            edge_point[self.local_edges] // self.world, meas[self.local_edges])

    def points_all(self):
        """All landmarks after optimisation (each rank contributes its shard; sum all-reduce of a zero-filled array)."""
        import torch
        out = np.zeros((self._P_all, 3))
        out[self.local_points] = self.points()
        t = torch.from_numpy(out)
        if self.dist.get_backend() == "nccl":
            d = t.cuda(); self.dist.all_reduce(d); out = d.cpu().numpy()
        else:
            self.dist.all_reduce(t)
        return out

    def comm_detail(self):
        """Bytes received by kind of exchange, the replicated part of the solves (ms, count), whether the segment-wise solve is on."""
        o = np.zeros(8)
        self.ctx.check(_lib.lib().bh_ba_get_comm_detail(self.h, _lib.ptr(o)))
        return dict(allreduce_bytes=int(o[0]), reduce_scatter_bytes=int(o[1]), all_gather_bytes=int(o[2]), replicated_ms=float(o[3]),
                    solves=int(o[4]), segment_solve=bool(o[5]), blocks_this_rank=int(o[6]), blocks_per_rank=int(o[7]))

    def comm(self):
        ms = C.c_double(0); nb = C.c_longlong(0)
        self.ctx.check(_lib.lib().bh_ba_get_comm(self.h, C.byref(ms), C.byref(nb)))
        return ms.value, nb.value
