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


def ba_optimize(R, t, cam_fixed, pts, ecam, ept, meas, K, iters, huber_delta, verbose_errors=True, pt_fixed=None):
    """g2o-style LM/Schur BA.  R [C][3][3], t [C][3] camera-in-world.  Returns dict (R, t, pts, chi2, log, iters_done, rc)."""
    R = np.array(R, dtype=np.float64).reshape(-1, 9).copy(); t = np.array(t, dtype=np.float64).reshape(-1, 3).copy()
    pts = np.array(pts, dtype=np.float64).reshape(-1, 3).copy()
    Cn, Pn = len(R), len(pts)
    cam_fixed = np.ascontiguousarray(cam_fixed, dtype=np.uint8)
    ecam = np.ascontiguousarray(ecam, dtype=np.int32); ept = np.ascontiguousarray(ept, dtype=np.int32)
    meas = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 2)
    En = len(ecam)
    K = np.ascontiguousarray(K, dtype=np.float64)
    chi2 = np.zeros(max(En, 1), np.float64); log = np.zeros((max(iters, 1), 4), np.float64)
    done = C.c_int(0)
    pf = None if pt_fixed is None else np.ascontiguousarray(pt_fixed, dtype=np.uint8)
    rc = lib().orc_ba_optimize(Cn, _p(R), _p(t), _p(cam_fixed), Pn, _p(pts), _p(pf), En, _p(ecam), _p(ept), _p(meas),
                               _p(K), int(iters), C.c_double(huber_delta), int(bool(verbose_errors)), _p(chi2), _p(log),
                               C.byref(done))
    return dict(R=R.reshape(-1, 3, 3), t=t, pts=pts, chi2=chi2[:En], log=log[:done.value], iters_done=done.value, rc=rc)


def pose_only_ba(pix, pts, K, iters=10, huber_delta=np.sqrt(7.815), chi2_thr=7.815):
    pix = np.ascontiguousarray(pix, dtype=np.float64).reshape(-1, 2)
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
    n = len(pix)
    K = np.ascontiguousarray(K, dtype=np.float64)
    R = np.zeros(9); t = np.zeros(3); chi2 = np.zeros(max(n, 1)); inl = np.zeros(max(n, 1), np.uint8)
    rc = lib().orc_pose_only_ba(_p(pix), _p(pts), n, _p(K), int(iters), C.c_double(huber_delta), C.c_double(chi2_thr),
                                _p(R), _p(t), _p(chi2), _p(inl))
    return dict(R=R.reshape(3, 3), t=t, chi2=chi2[:n], inlier=inl[:n].astype(bool), rc=rc)


# ---- single-stage helpers for tests -----------------------------------------------------------------------------
def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    out = np.zeros(bgr.shape[:2], np.uint8)
    lib().orc_bgr2gray(_p(bgr), out.size, _p(out))
    return out


def resize_linear_exact(img, dw, dh):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_exact(_p(img), img.shape[1], img.shape[0], _p(out), dw, dh)
    return out


def fast_score_map(img, thr):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros_like(img)
    lib().orc_fast_score_map(_p(img), img.shape[1], img.shape[0], int(thr), _p(out))
    return out


def fast_nms(score):
    score = np.ascontiguousarray(score, dtype=np.uint8)
    out = np.zeros_like(score)
    lib().orc_fast_nms(_p(score), score.shape[1], score.shape[0], _p(out))
    return out


def gauss_blur7(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros_like(img)
    lib().orc_gauss_blur7(_p(img), img.shape[1], img.shape[0], _p(out))
    return out


def harris_at(img, x, y):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    f = lib().orc_harris_at; f.restype = C.c_float
    return np.float32(f(_p(img), img.shape[1], int(x), int(y)))


def ic_angle_at(img, x, y):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    f = lib().orc_ic_angle_at; f.restype = C.c_float
    return np.float32(f(_p(img), img.shape[1], int(x), int(y)))


def umax_table():
    out = np.zeros(16, np.int32)
    lib().orc_umax_table(_p(out))
    return out


def brief_at(blur, x, y, angle_deg):
    blur = np.ascontiguousarray(blur, dtype=np.uint8)
    out = np.zeros(32, np.uint8)
    lib().orc_brief_at(_p(blur), blur.shape[1], int(x), int(y), C.c_float(angle_deg), _p(out))
    return out


def pattern():
    f = lib().orc_pattern; f.restype = C.POINTER(C.c_byte * 1024)
    return np.frombuffer(f().contents, dtype=np.int8).copy()


def fast_atan2(y, x):
    return np.float32(lib().orc_fast_atan2(C.c_float(y), C.c_float(x)))


def visibility_gate(pts, normals, R, t, f, cx, cy, width, height, see, cos_thr=np.cos(np.pi / 3)):
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3); normals = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
    R = np.ascontiguousarray(R, dtype=np.float64); t = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
    see = np.ascontiguousarray(see, dtype=np.float64).reshape(3)
    idx = np.zeros(max(len(pts), 1), np.int32)
    n = lib().orc_visibility_gate(_p(pts), _p(normals), len(pts), _p(R), _p(t), C.c_double(f), C.c_double(cx), C.c_double(cy),
                                  int(width), int(height), _p(see), C.c_double(cos_thr), _p(idx))
    return idx[:n].copy()


def bow_query_l1(q_words, q_vals, start, words, vals, max_results=1):
    q_words = np.ascontiguousarray(q_words, dtype=np.uint32); q_vals = np.ascontiguousarray(q_vals, dtype=np.float64)
    start = np.ascontiguousarray(start, dtype=np.int64); words = np.ascontiguousarray(words, dtype=np.uint32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    ids = np.zeros(max(max_results, 1), np.int32); sc = np.zeros(max(max_results, 1), np.float64)
    n = lib().orc_bow_query_l1(_p(q_words), _p(q_vals), len(q_words), _p(start), _p(words), _p(vals), len(start) - 1,
                               int(max_results), _p(ids), _p(sc))
    return ids[:n].copy(), sc[:n].copy()


def bow_query_l1_subset(q_words, q_vals, start, words, vals, subset, max_results):
    q_words = np.ascontiguousarray(q_words, dtype=np.uint32); q_vals = np.ascontiguousarray(q_vals, dtype=np.float64)
    start = np.ascontiguousarray(start, dtype=np.int64); words = np.ascontiguousarray(words, dtype=np.uint32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    allowed = np.zeros(len(start) - 1, np.uint8); allowed[np.asarray(subset, dtype=np.int64)] = 1
    ids = np.zeros(max(max_results, 1), np.int32); sc = np.zeros(max(max_results, 1), np.float64)
    n = lib().orc_bow_query_l1_subset(_p(q_words), _p(q_vals), len(q_words), _p(start), _p(words), _p(vals), len(start) - 1,
                                      _p(allowed), int(max_results), _p(ids), _p(sc))
    return ids[:n].copy(), sc[:n].copy()


def bow_compare_l1(start, words, vals, i, j):
    start = np.ascontiguousarray(start, dtype=np.int64); words = np.ascontiguousarray(words, dtype=np.uint32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    cnt = C.c_int(0); score = C.c_double(0.0)
    lib().orc_bow_compare_l1(_p(start), _p(words), _p(vals), int(i), int(j), C.byref(cnt), C.byref(score))
    return cnt.value, score.value


def rodrigues(v):
    """cv2.Rodrigues semantics: 3-vector -> 3x3 matrix, or 3x3 matrix -> 3-vector."""
    v = np.ascontiguousarray(v, dtype=np.float64)
    if v.size == 3:
        R = np.zeros(9); lib().orc_rodrigues_vec2mat(_p(v.reshape(3)), _p(R)); return R.reshape(3, 3)
    r = np.zeros(3); lib().orc_rodrigues_mat2vec(_p(v.reshape(9)), _p(r)); return r


def ps_qr_solve_6x4(A, b):
    """pnp_small.h: least-squares solution of a 6 x 4 system by Householder QR (EPnP's Gauss-Newton step)."""
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(6, 4); b = np.ascontiguousarray(b, dtype=np.float64).reshape(6)
    x = np.zeros(4); lib().orc_ps_qr_solve_6x4(_p(A), _p(b), _p(x)); return x


def ps_chol6_solve(A, b):
    """pnp_small.h: 6 x 6 SPD solve by Cholesky (lower triangle read).  Returns (ok, x); x = 0 when a pivot fails."""
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(6, 6); b = np.ascontiguousarray(b, dtype=np.float64).reshape(6)
    x = np.zeros(6); ok = lib().orc_ps_chol6_solve(_p(A), _p(b), _p(x)); return bool(ok), x


def ba_cube_mode(two_mul=None):
    """(2 rho - 1)^3 of the Levenberg lambda update: 0 = pow() as g2o (default), 1 = two rounded multiplications.  Returns the old mode."""
    return int(lib().orc_ba_cube_mode(-1 if two_mul is None else int(bool(two_mul))))


def solve_pnp_ransac(obj, img, K4, iterations=100, reproj_error=8.0, confidence=0.99):
    obj = np.ascontiguousarray(obj, dtype=np.float64).reshape(-1, 3); img = np.ascontiguousarray(img, dtype=np.float64).reshape(-1, 2)
    n = len(obj)
    K4 = np.ascontiguousarray(K4, dtype=np.float64)
    rvec = np.zeros(3); tvec = np.zeros(3); inl = np.zeros(max(n, 1), np.int32); ni = C.c_int(0)
    ok = lib().orc_solve_pnp_ransac(_p(obj), _p(img), n, _p(K4), int(iterations), C.c_double(reproj_error), C.c_double(confidence),
                                    _p(rvec), _p(tvec), _p(inl), C.byref(ni))
    return bool(ok), rvec, tvec, inl[:ni.value].copy()


def p3_roots(A):
    """positive real roots of the quartic A[0] v^4 + .. + A[4] as p3p_small.h finds them (sign scan + bisection)"""
    A = np.ascontiguousarray(A, dtype=np.float64); v = np.zeros(4)
    n = lib().orc_p3_roots(_p(A), _p(v))
    return v[:n].copy()


def p3_solve3(X, f):
    """up to four poses (R, t) with s_i f_i = R X_i + t for three correspondences (p3p_small.h)"""
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(3, 3); f = np.ascontiguousarray(f, dtype=np.float64).reshape(3, 3)
    Rs = np.zeros((4, 3, 3)); ts = np.zeros((4, 3))
    n = lib().orc_p3_solve3(_p(X), _p(f), _p(Rs), _p(ts))
    return Rs[:n].copy(), ts[:n].copy()


def p3_solve4(obj, img, K4):
    """solvePnPRansac's four-point model (p3p_small.h): (ok, R, t)"""
    obj = np.ascontiguousarray(obj, dtype=np.float64).reshape(4, 3); img = np.ascontiguousarray(img, dtype=np.float64).reshape(4, 2)
    K4 = np.ascontiguousarray(K4, dtype=np.float64); R = np.zeros((3, 3)); t = np.zeros(3)
    ok = lib().orc_p3_solve4(_p(obj), _p(img), _p(K4), _p(R), _p(t))
    return bool(ok), R, t


def mp_add_observation(obs, n_obs, normal, feat, new_desc, new_n):
    """MapPoint.add_observation, slam/nodes.py:142-156, line by line in numpy as the reference runs it (dbow.distance = Hamming distance
    of the two descriptors).  obs: list of u8[32] descriptors.  Returns (obs, n_obs, normal, feat)."""
    n = np.array(new_n, dtype=np.float64)
    n /= np.linalg.norm(n)
    n_obs += 1
    obs = list(obs)
    if len(obs) < 10:
        obs.append(np.array(new_desc, dtype=np.uint8))
        dists = np.zeros((len(obs), len(obs)), dtype=np.uint8)
        for i1, f1 in enumerate(obs):
            for i2, f2 in enumerate(obs):
                if i1 < i2:
                    dists[i2, i1] = dists[i1, i2] = np.uint8(int(np.unpackbits(np.bitwise_xor(f1, f2)).sum()) & 255)
        feat = obs[np.argmin(np.median(dists, axis=0))]
    normal = (n + (n_obs - 1) * np.asarray(normal, dtype=np.float64)) / n_obs
    return obs, n_obs, normal, feat


def align_se3_ransac(P, Q, iterations=200, thr=0.05, confidence=0.99, min_inliers=15):
    """q ~ R p + t by 3-point RANSAC + Kabsch (align_oracle.c).  Returns (status, R, t, mask, n_inliers)."""
    P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, 3); Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(-1, 3)
    n = len(P)
    R = np.zeros((3, 3)); t = np.zeros(3); mask = np.zeros(max(n, 1), np.uint8); ni = C.c_int(0)
    st = lib().orc_align_se3_ransac(_p(P), _p(Q), n, int(iterations), C.c_double(thr), C.c_double(confidence), int(min_inliers), _p(R), _p(t), _p(mask),
                                    C.byref(ni))
    return int(st), R, t, mask[:n].astype(bool), ni.value


def se_kabsch(H, pbar, qbar):
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(9); pbar = np.ascontiguousarray(pbar, dtype=np.float64); qbar = np.ascontiguousarray(qbar, dtype=np.float64)
    R = np.zeros((3, 3)); t = np.zeros(3)
    ok = lib().orc_se_kabsch(_p(H), _p(pbar), _p(qbar), _p(R), _p(t))
    return bool(ok), R, t


def voc_transform(voc, desc):
    """voc: dict(child_start, child_idx, node_desc, node_weight, node_word) -> (words u32 ascending, vals f64)."""
    desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
    cs = np.ascontiguousarray(voc["child_start"], dtype=np.int32); ci = np.ascontiguousarray(voc["child_idx"], dtype=np.int32)
    nd = np.ascontiguousarray(voc["node_desc"], dtype=np.uint8); nw = np.ascontiguousarray(voc["node_weight"], dtype=np.float64)
    ni = np.ascontiguousarray(voc["node_word"], dtype=np.int32)
    words = np.zeros(max(len(desc), 1), np.uint32); vals = np.zeros(max(len(desc), 1), np.float64)
    k = lib().orc_voc_transform(_p(cs), _p(ci), _p(nd), _p(nw), _p(ni), _p(desc), len(desc), _p(words), _p(vals))
    return words[:k].copy(), vals[:k].copy()


def medoid_index(desc):
    desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
    return int(lib().orc_medoid_index(_p(desc), len(desc)))


def window_problem(nbr_start, nbr_id, nbr_common, obs_start, obs_mp, obs_px, kf, min_common=15):
    """Pure-Python restatement of slam/covisibility_graph.py:51-78 (get_local_map, flag_inner_outer=True, flag_with_input_kf=False)
    and :188-218 (the walk of optimize_local up to the add_* calls), with every Python set enumerated in ascending id.
    TEST INFRASTRUCTURE.  PARITY UNPINNED (the reference holds no fixture for it; this follows its source line by line)."""
    from collections import Counter
    nbrs = lambda k: [(int(nbr_id[e]), int(nbr_common[e])) for e in range(nbr_start[k], nbr_start[k + 1])]
    common_with_kf = {i: c for i, c in nbrs(kf)}                                     # kf2kf_num_common_mps[(i, kf)], defaultdict -> 0
    inner, outer = {kf}, set()
    for kf1, c1 in nbrs(kf):                                                          # :64
        if c1 >= min_common:                                                          # :65
            inner.add(kf1)
            for kf2, _ in nbrs(kf1):                                                  # :67
                if kf2 != kf and common_with_kf.get(kf2, 0) >= min_common:            # :68-69
                    outer.add(kf2)
    inner.remove(kf)                                                                  # :72-73
    outer = outer - inner                                                             # :76
    inner, outer = sorted(inner), sorted(outer)
    if not outer and not inner:                                                       # :191-194
        return dict(win_kf=np.zeros(0, np.int32), win_fixed=np.zeros(0, bool), n_inner=0, n_outer=0, pt_mp=np.zeros(0, np.int32),
                    edge_pose=np.zeros(0, np.int32), edge_point=np.zeros(0, np.int32), edge_px=np.zeros((0, 2), np.int64))
    chain = [kf] + inner + outer                                                      # :203
    fixed = [len(outer) == 0] + [False] * len(inner) + [True] * len(outer)            # :195-201
    counter = Counter(); pairs = []
    for w, i in enumerate(chain):                                                     # :205-208
        mps = [int(m) for m in obs_mp[obs_start[i]:obs_start[i + 1]]]
        counter.update(mps)
        pairs.extend((w, m, obs_start[i] + j) for j, m in enumerate(mps))
    slot, pt_mp, e_pose, e_pt, e_px = {}, [], [], [], []
    for w, m, src in pairs:                                                           # :211-216
        if counter[m] > 1:
            if m not in slot:
                slot[m] = len(pt_mp); pt_mp.append(m)
            e_pose.append(w); e_pt.append(slot[m]); e_px.append(obs_px[src])
    return dict(win_kf=np.array(chain, np.int32), win_fixed=np.array(fixed, bool), n_inner=len(inner), n_outer=len(outer),
                pt_mp=np.array(pt_mp, np.int32), edge_pose=np.array(e_pose, np.int32), edge_point=np.array(e_pt, np.int32),
                edge_px=np.array(e_px, np.int64).reshape(-1, 2))


def loop_fuse_pairs(feat_a, pt_a, nobs_a, feat_b, pt_b, nobs_b, R, t, d_max=30, radius=0.05):
    """The project's specification of GlobalMapManager._loop_fusion (slam/loop_closing.py:21-26; the reference body is `pass`: parity unpinned
    by construction): cross-check Hamming matches of the two sides' map-point descriptors with distance < d_max whose 3-D points agree under
    a ~ R b + t to within `radius`; the point with at least as many observations survives.  Returns (pair_a, pair_b, keep_a)."""
    feat_a = np.ascontiguousarray(feat_a, dtype=np.uint8).reshape(-1, 32); feat_b = np.ascontiguousarray(feat_b, dtype=np.uint8).reshape(-1, 32)
    pt_a = np.asarray(pt_a, dtype=np.float64).reshape(-1, 3); pt_b = np.asarray(pt_b, dtype=np.float64).reshape(-1, 3)
    R = np.asarray(R, dtype=np.float64).reshape(3, 3); t = np.asarray(t, dtype=np.float64).reshape(3)
    if len(feat_a) == 0 or len(feat_b) == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, bool)
    qi, ti, d = match_crosscheck(feat_a, feat_b)
    pa, pb, ka = [], [], []
    for a, b, dist in zip(qi, ti, d):
        if dist >= d_max:
            continue
        x = pt_b[b]
        diff = [pt_a[a][k] - (((R[k, 0] * x[0]) + (R[k, 1] * x[1])) + (R[k, 2] * x[2]) + t[k]) for k in range(3)]
        if ((diff[0] * diff[0]) + (diff[1] * diff[1])) + (diff[2] * diff[2]) <= radius * radius:
            pa.append(a); pb.append(b); ka.append(nobs_a[a] >= nobs_b[b])
    return np.array(pa, np.int32), np.array(pb, np.int32), np.array(ka, bool)


# ---- essential-graph optimisation: the project's specification of GlobalMapManager._essential_graph_optimization (slam/loop_closing.py:28-29;
# the reference body is `pass`: parity unpinned by construction).  Poses (R_i, t_i) camera-in-world, edges Z_ij ~ T_i^-1 T_j, residual
# r = (t_E, log R_E) of E = Z^-1 T_i^-1 T_j, update T <- T (+) d: R <- R Exp(d_w), t <- t + R d_v.
def _so3_log(R):
    c = min(1.0, max(-1.0, 0.5 * (np.trace(R) - 1.0)))
    th = np.arccos(c)
    k = 0.5 + th * th / 12.0 if th < 1e-7 else th / (2.0 * np.sin(th))
    return k * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])


def _hat(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def _so3_exp(w):
    th2 = float(w @ w); th = np.sqrt(th2)
    a = 1.0 - th2 / 6.0 if th < 1e-7 else np.sin(th) / th
    b = 0.5 - th2 / 24.0 if th < 1e-7 else (1.0 - np.cos(th)) / th2
    W = _hat(w)
    return np.eye(3) + a * W + b * (W @ W)


def _jr_inv(w):
    th2 = float(w @ w); th = np.sqrt(th2)
    c = 1.0 / 12.0 + th2 / 720.0 if th < 1e-4 else 1.0 / th2 - (1.0 + np.cos(th)) / (2.0 * th * np.sin(th))
    W = _hat(w)
    return np.eye(3) + 0.5 * W + c * (W @ W)


def pose_graph_edge(Ri, ti, Rj, tj, Zr, Zt, jac=True):
    """Residual r (6) and the Jacobians A = dr/d(d_i), B = dr/d(d_j) (6 x 6, columns (d_v, d_w)) of one edge."""
    RE = Zr.T @ Ri.T @ Rj
    a = Ri.T @ (tj - ti)
    tE = Zr.T @ (a - Zt)
    ph = _so3_log(RE)
    r = np.concatenate([tE, ph])
    if not jac:
        return r, None, None
    Ji = _jr_inv(ph)
    A = np.zeros((6, 6)); B = np.zeros((6, 6))
    A[:3, :3] = -Zr.T; A[:3, 3:] = Zr.T @ _hat(a); A[3:, 3:] = -Ji @ RE.T @ Zr.T
    B[:3, :3] = RE; B[3:, 3:] = Ji
    return r, A, B


def pose_graph_optimize(R, t, fixed, ei, ej, Zr, Zt, w, iters=10):
    """Gauss-Newton over the free poses, every step the exact solution of (J'WJ + 1e-9 I) d = -J'W r (dense, numpy), kept only if it lowers
    the cost.  Returns (R, t, cost_log, steps_kept)."""
    R = np.array(R, dtype=np.float64).reshape(-1, 3, 3); t = np.array(t, dtype=np.float64).reshape(-1, 3)
    K, E = len(R), len(ei)
    free = [n for n in range(K) if not fixed[n]]
    col = {n: 6 * k for k, n in enumerate(free)}

    def cost_of(Rc, tc):
        return sum(w[e] * float(np.sum(pose_graph_edge(Rc[ei[e]], tc[ei[e]], Rc[ej[e]], tc[ej[e]], Zr[e], Zt[e], False)[0] ** 2)) for e in range(E))
    log = [cost_of(R, t)]
    kept = 0
    for it in range(iters):
        H = np.zeros((6 * len(free), 6 * len(free))); g = np.zeros(6 * len(free))
        for e in range(E):
            i, j = int(ei[e]), int(ej[e])
            r, A, B = pose_graph_edge(R[i], t[i], R[j], t[j], Zr[e], Zt[e])
            for n, J in ((i, A), (j, B)):
                if n in col:
                    g[col[n]:col[n] + 6] -= w[e] * (J.T @ r)
                    for m, J2 in ((i, A), (j, B)):
                        if m in col:
                            H[col[n]:col[n] + 6, col[m]:col[m] + 6] += w[e] * (J.T @ J2)
        d = np.linalg.solve(H + 1e-9 * np.eye(len(H)), g) if len(H) else np.zeros(0)
        Rn, tn = R.copy(), t.copy()
        for n in free:
            dv, dw = d[col[n]:col[n] + 3], d[col[n] + 3:col[n] + 6]
            tn[n] = t[n] + R[n] @ dv; Rn[n] = R[n] @ _so3_exp(dw)
        c = cost_of(Rn, tn)
        if not c < log[-1]:
            break
        R, t = Rn, tn
        log.append(c); kept += 1
    return R, t, np.array(log), kept

