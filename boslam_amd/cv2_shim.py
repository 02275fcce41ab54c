"""Call-compatible stand-ins for the cv2 entry points the reference's hot path uses, backed by libboslam_hip.so.

  cv2.ORB_create(**orb_params).detectAndCompute(frame, mask=...)      camera.py:62,141
  cv2.BFMatcher_create(cv2.NORM_HAMMING, crossCheck=True).match(q, t)  slam/tracking.py:45,56,121,
                                                                       slam/local_mapping.py:21,
                                                                       slam/covisibility_graph.py:34,
                                                                       experiments/pnp_two_way_tracking.py:11,42
A maintainer swaps `import cv2` for `from boslam_amd import cv2_shim as cv2` at those sites.
Keypoints come back in the library's canonical order (level-major, raster inside a level) — OpenCV's own order inside
a level is unspecified (it is whatever std::nth_element leaves)."""
import numpy as np
from . import _lib, frontend

NORM_HAMMING = 6
ORB_HARRIS_SCORE = 0
ORB_FAST_SCORE = 1


class KeyPoint:
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, x, y, size, angle=-1.0, response=0.0, octave=0, class_id=-1):
        self.pt = (float(x), float(y)); self.size = float(size); self.angle = float(angle)
        self.response = float(response); self.octave = int(octave); self.class_id = int(class_id)


class DMatch:
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx, trainIdx, distance, imgIdx=0):
        self.queryIdx = int(queryIdx); self.trainIdx = int(trainIdx); self.imgIdx = int(imgIdx)
        self.distance = float(distance)

    def __repr__(self):
        return f"DMatch({self.queryIdx}, {self.trainIdx}, {self.distance})"


class _ORB:
    def __init__(self, params, ctx=None):
        self._params = params
        self._ctx = ctx

    def detectAndCompute(self, image, mask=None):
        ctx = self._ctx or _lib.default_context()
        image = np.ascontiguousarray(image, dtype=np.uint8)
        if image.ndim not in (2, 3):
            raise ValueError("image must be HxW or HxWx3 uint8")
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)[None]
        kps, desc, n = frontend.orb_detect_describe(ctx, image[None], mask=m, params=self._params)
        n = int(n[0])
        if n == 0:
            return [], None                                       # experiments/pnp_two_way_tracking.py:38 tests `des is None`
        k = kps[0, :n]
        out = [KeyPoint(k["x"][i], k["y"][i], k["size"][i], k["angle"][i], k["response"][i], k["octave"][i]) for i in range(n)]
        return out, desc[0, :n].copy()

    def detect_batch(self, images, depth=None, mask=None, max_kp=None):
        """Batched form (no cv2 equivalent): arrays in, arrays out — see frontend.orb_detect_describe."""
        ctx = self._ctx or _lib.default_context()
        return frontend.orb_detect_describe(ctx, images, mask=mask, depth=depth, params=self._params, max_kp=max_kp)


def ORB_create(nfeatures=500, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2,
               scoreType=ORB_HARRIS_SCORE, patchSize=31, fastThreshold=20, ctx=None):
    if WTA_K != 2 or scoreType != ORB_HARRIS_SCORE:
        raise ValueError("only WTA_K=2 / HARRIS_SCORE (the defaults the reference relies on, camera.py:49-56)")
    return _ORB(frontend.make_params(nfeatures=nfeatures, scaleFactor=scaleFactor, nlevels=nlevels,
                                     edgeThreshold=edgeThreshold, firstLevel=firstLevel, patchSize=patchSize,
                                     fastThreshold=fastThreshold), ctx)


class _BFMatcher:
    def __init__(self, ctx=None):
        self._ctx = ctx

    def match(self, queryDescriptors, trainDescriptors):
        if queryDescriptors is None or trainDescriptors is None:
            return []
        q = np.ascontiguousarray(queryDescriptors, dtype=np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(trainDescriptors, dtype=np.uint8).reshape(-1, 32)
        if len(q) == 0 or len(t) == 0:
            return []                                             # callers guard len(matches) == 0 (tracking.py:58)
        ctx = self._ctx or _lib.default_context()
        qi, ti, d, nm = frontend.match_pairs(ctx, q[None], [len(q)], t[None], [len(t)])
        return [DMatch(qi[0, i], ti[0, i], d[0, i]) for i in range(int(nm[0]))]


def BFMatcher_create(normType=NORM_HAMMING, crossCheck=False, ctx=None):
    if normType != NORM_HAMMING or not crossCheck:
        raise ValueError("only BFMatcher(NORM_HAMMING, crossCheck=True) — the one configuration the reference creates")
    return _BFMatcher(ctx)


# ---- calib3d stand-ins: experiments/pnp_two_way_tracking.py:57-61,68-69 ------------------------------------------------
SOLVEPNP_ITERATIVE = 0
SOLVEPNP_EPNP = 1


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs=None, rvec=None, tvec=None, useExtrinsicGuess=False,
                   iterationsCount=100, reprojectionError=8.0, confidence=0.99, inliers=None, flags=SOLVEPNP_ITERATIVE, ctx=None):
    """-> (retval, rvec (3,1), tvec (3,1), inliers (k,1) int32 | None), like cv2.  The final pose is the reprojection-error
    minimiser over the RANSAC inliers for both ITERATIVE and EPNP flags.

    Exactly four points (round 5): OpenCV 3.x switches its minimal solver to P3P when n == 4 - one hypothesis from all four points (the
    first three give up to four poses, the fourth picks one), accepted only when all four are inliers - and so does this (p3p_small.h: the
    same construction restated from the classical derivation, OpenCV's p3p.cpp not being available; parity unpinned like the rest of the
    path).  The reference never gets there: its only live call site needs >= 10 matches first (experiments/pnp_two_way_tracking.py:45,57-61).
    Any iterationsCount is accepted."""
    import ctypes as C
    if distCoeffs is not None and np.any(np.asarray(distCoeffs) != 0):
        raise ValueError("non-zero distortion is not supported (the reference always passes zeros, camera.py:82)")
    if useExtrinsicGuess:
        raise ValueError("useExtrinsicGuess is not supported")
    obj = np.ascontiguousarray(objectPoints, dtype=np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(imagePoints, dtype=np.float64).reshape(-1, 2)
    n = len(obj)
    if n < 4 or len(img) != n:
        raise ValueError("solvePnPRansac needs >= 4 point pairs")                 # cv2 raises cv2.error here
    Km = np.asarray(cameraMatrix, dtype=np.float64)
    K4 = np.array([Km[0, 0], Km[1, 1], Km[0, 2], Km[1, 2]], dtype=np.float64)
    ctx = ctx or _lib.default_context()
    rv = np.zeros((1, 3)); tv = np.zeros((1, 3)); inl = np.zeros((1, n), np.int32); ni = np.zeros(1, np.int32); ok = np.zeros(1, np.int32)
    nn = np.array([n], np.int32)
    ctx.check(_lib.lib().bh_solve_pnp_ransac(ctx.h, _lib.ptr(obj), _lib.ptr(img), _lib.ptr(nn), 1, n, _lib.ptr(K4), int(iterationsCount),
                                             float(reprojectionError), float(confidence), _lib.ptr(rv), _lib.ptr(tv), _lib.ptr(inl),
                                             _lib.ptr(ni), _lib.ptr(ok), _lib.BH_MEM_HOST))
    if not ok[0]:
        return False, rv.reshape(3, 1), tv.reshape(3, 1), None
    return True, rv.reshape(3, 1).copy(), tv.reshape(3, 1).copy(), inl[0, :ni[0]].reshape(-1, 1).copy()


def Rodrigues(src):
    """-> (dst, jacobian=None): 3-vector -> 3x3 matrix or 3x3 -> (3,1) vector."""
    v = np.ascontiguousarray(src, dtype=np.float64)
    if v.size == 3:
        out = np.zeros((3, 3))
        rc = _lib.lib().bh_rodrigues(_lib.ptr(v.reshape(3)), 3, _lib.ptr(out))
    elif v.size == 9:
        out = np.zeros((3, 1))
        rc = _lib.lib().bh_rodrigues(_lib.ptr(v.reshape(9)), 9, _lib.ptr(out))
    else:
        raise ValueError("Rodrigues expects 3 or 9 elements")
    if rc != 0:
        raise ValueError("bh_rodrigues failed")
    return out, None
