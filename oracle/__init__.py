"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under boslam_amd/ may import this package (tests/test_no_oracle_in_product.py enforces it).
PARITY UNPINNED vs OpenCV/g2o — see the headers of the C files.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scaleFactor", C.c_float), ("nlevels", C.c_int),
                ("edgeThreshold", C.c_int), ("firstLevel", C.c_int), ("patchSize", C.c_int),
                ("fastThreshold", C.c_int)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_orb_geometry.restype = C.c_size_t
        _LIB.orc_fast_atan2.restype = C.c_float
        _LIB.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def orb_params(nfeatures=600, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, patchSize=31,
               fastThreshold=20):
    return OrbParams(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, patchSize, fastThreshold)


def orb_geometry(H, W, params):
    L = params.nlevels
    w = np.zeros(L, np.int32); h = np.zeros(L, np.int32)
    off = np.zeros(L, np.uint64); nf = np.zeros(L, np.int32)
    total = lib().orc_orb_geometry(H, W, C.byref(params), _p(w), _p(h), _p(off), _p(nf))
    return int(total), w, h, off.astype(np.int64), nf


def orb_detect_describe(img, mask=None, params=None, max_kp=None, debug=False):
    """One frame. Returns (kps structured array, desc u8[N][32], rc[, dbg dict])."""
    params = params or orb_params()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
    max_kp = max_kp or (params.nfeatures * 2 + 256)
    kps = np.zeros(max_kp, KP_DTYPE)
    desc = np.zeros((max_kp, 32), np.uint8)
    n = C.c_int(0)
    dbg = None
    if debug:
        total, w, h, off, nf = orb_geometry(H, W, params)
        dbg = {k: np.zeros(total, np.uint8) for k in ("pyr", "mask", "nms", "blur")}
        dbg.update(w=w, h=h, off=off, nfeat=nf)
        rc = lib().orc_orb_detect_describe_dbg(_p(img), ch, _p(mask), H, W, C.byref(params), max_kp, _p(kps),
                                               _p(desc), C.byref(n), _p(dbg["pyr"]), _p(dbg["mask"]),
                                               _p(dbg["nms"]), _p(dbg["blur"]))
    else:
        rc = lib().orc_orb_detect_describe(_p(img), ch, _p(mask), H, W, C.byref(params), max_kp, _p(kps),
                                           _p(desc), C.byref(n))
    out = (kps[:n.value].copy(), desc[:n.value].copy(), rc)
    return out + (dbg,) if debug else out


def orb_batch(imgs, depth, params, max_kp):
    """Batched frames (OpenMP over frames). imgs u8[B][H][W][C], depth u16[B][H][W] or None."""
    imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
    B, H, W = imgs.shape[:3]
    ch = 1 if imgs.ndim == 3 else imgs.shape[3]
    if depth is not None:
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
    kps = np.zeros((B, max_kp), KP_DTYPE)
    desc = np.zeros((B, max_kp, 32), np.uint8)
    n = np.zeros(B, np.int32)
    rc = lib().orc_orb_batch(_p(imgs), ch, _p(depth), B, H, W, C.byref(params), max_kp, _p(kps), _p(desc), _p(n))
    return kps, desc, n, rc


def match_crosscheck(q, t):
    q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, dtype=np.uint8).reshape(-1, 32)
    m = max(1, min(len(q), len(t)))
    qi = np.zeros(m, np.int32); ti = np.zeros(m, np.int32); d = np.zeros(m, np.int32)
    n = lib().orc_match_crosscheck(_p(q), len(q), _p(t), len(t), _p(qi), _p(ti), _p(d))
    return qi[:n].copy(), ti[:n].copy(), d[:n].copy()


def match_consecutive(desc, n, max_kp):
    desc = np.ascontiguousarray(desc, dtype=np.uint8)
    n = np.ascontiguousarray(n, dtype=np.int32)
    B = len(n)
    qi = np.zeros((B, max_kp), np.int32); ti = np.zeros((B, max_kp), np.int32)
    d = np.zeros((B, max_kp), np.int32); nm = np.zeros(B, np.int32)
    lib().orc_match_consecutive(_p(desc), _p(n), B, max_kp, _p(qi), _p(ti), _p(d), _p(nm))
    return qi, ti, d, nm


def kp_to_int(kps):
    kps = np.ascontiguousarray(kps)
    xy = np.zeros((len(kps), 2), np.int64)
    lib().orc_kp_to_int(_p(kps), len(kps), _p(xy))
    return xy


def backproject(depth, xy, fx, fy, ppx, ppy):
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    xy = np.ascontiguousarray(xy, dtype=np.int64)
    out = np.zeros((len(xy), 3), np.float64)
    lib().orc_backproject(_p(depth), depth.shape[1], _p(xy), len(xy), C.c_double(fx), C.c_double(fy),
                          C.c_double(ppx), C.c_double(ppy), _p(out))
    return out


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))
