"""ctypes binding of libboslam_hip.so (include/boslam_hip.h).  No CPU fallback: if the library or a HIP device is
missing every entry point raises."""
import ctypes as C
import os
import threading
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BH_LIB_PATH") or os.path.join(_HERE, "csrc", "libboslam_hip.so")    # BH_LIB_PATH: A/B builds of the library

BH_OK, BH_ERR_INVALID, BH_ERR_HIP, BH_ERR_CAPACITY, BH_ERR_NOMEM, BH_ERR_NUMERIC = 0, -1, -2, -3, -4, -5
BH_MEM_HOST, BH_MEM_DEVICE = 0, 1


class BhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"boslam_hip error {code}: {msg}")
        self.code = code


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scaleFactor", C.c_float), ("nlevels", C.c_int),
                ("edgeThreshold", C.c_int), ("firstLevel", C.c_int), ("patchSize", C.c_int),
                ("fastThreshold", C.c_int)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])

_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
_SIGS = {
    "bh_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "bh_ctx_destroy": (None, [_vp]),
    "bh_last_error": (C.c_char_p, [_vp]),
    "bh_ctx_set_stream": (_i, [_vp, _vp]),
    "bh_ctx_synchronize": (_i, [_vp]),
    "bh_ctx_status": (_i, [_vp]),
    "bh_version": (C.c_char_p, []),
    "bh_ctx_timing": (_i, [_vp, _i]),
    "bh_ctx_timing_read": (_i, [_vp, _i, _vp, _vp, _vp, C.POINTER(_i)]),
    "bh_orb_detect_describe": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, C.POINTER(OrbParams), _i, _vp, _vp, _vp, _i]),
    "bh_orb_geometry": (_i, [_i, _i, C.POINTER(OrbParams), _vp, _vp, _vp, _vp, _vp, C.POINTER(_i)]),
    "bh_orb_debug_fetch": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "bh_kp_backproject": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _d, _d, _d, _d, _vp, _vp, _i]),
    "bh_match_hamming_crosscheck": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i]),
    "bh_medoid_descriptor": (_i, [_vp, _vp, _vp, _i, _vp, _i]),
    "bh_mp_add_observation": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "bh_pose_only_ba": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _i]),
    "bh_track_refkf": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i]),
    "bh_pose_graph_optimize": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, C.POINTER(_i), C.POINTER(_i)]),
    "bh_loop_fuse_pairs": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _d, _vp, _vp, _vp, _vp, _i, _i]),
    "bh_kf_add_map_points": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i), _i]),
    "bh_track_local_map": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _d, _i, _i, _i, _vp, _i, _d, _d,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i]),
    "bh_align_se3_ransac": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _d, _d, _i, _vp, _vp, _vp, _vp, _vp, _i]),
    "bh_ba_create": (_i, [_vp, _vp, C.POINTER(_vp)]),
    "bh_ba_destroy": (None, [_vp]),
    "bh_ba_set_problem": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i]),
    "bh_ba_optimize": (_i, [_vp, _i, _d, _i, C.POINTER(_i)]),
    "bh_ba_get_poses": (_i, [_vp, _vp, _vp, _i]),
    "bh_ba_get_points": (_i, [_vp, _vp, _i]),
    "bh_ba_get_chi2": (_i, [_vp, _vp, _i]),
    "bh_ba_get_log": (_i, [_vp, _vp, _i]),
    "bh_ba_solver": (_i, [_vp]),
    "bh_ba_debug_structure": (_i, [_vp, _i, _vp, C.c_longlong, C.POINTER(C.c_longlong)]),
    "bh_ba_batch_create": (_i, [_vp, _i, _vp, _vp]),
    "bh_ba_batch_set_problems": (_i, [_vp, _i, _vp, _vp]),
    "bh_ba_batch_destroy": (None, [_vp]),
    "bh_ba_batch_optimize": (_i, [_vp, _i, C.c_double, _i, _vp]),
    "bh_ctx_copy": (_i, [_vp, _vp, _vp, C.c_size_t, _i]),
    "bh_ba_debug_band_solve": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "bh_ba_window_problem": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "bh_ba_set_allreduce": (_i, [_vp, _vp, _vp, _i, _i]),
    "bh_ba_set_allreduce_ordered": (_i, [_vp, _i]),
    "bh_ba_get_comm": (_i, [_vp, C.POINTER(_d), C.POINTER(C.c_longlong)]),
    "bh_ba_get_comm_detail": (_i, [_vp, _vp]),
    "bh_solve_pnp_ransac": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _i]),
    "bh_rodrigues": (_i, [_vp, _i, _vp]),
    "bh_debug_pnp_trace": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bh_debug_f64_ops": (_i, [_vp, _vp, _vp, _i, _vp]),
    "bh_debug_p3p": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "bh_visibility_gate": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _d, _d, _d, _i, _i, _d, _vp, _vp, _i]),
    "bh_bow_create": (_i, [_vp, C.POINTER(_vp)]),
    "bh_bow_destroy": (None, [_vp]),
    "bh_bow_set_db": (_i, [_vp, C.c_longlong, _vp, _vp, _vp, _i]),
    "bh_bow_set_mode": (_i, [_vp, _i]),
    "bh_bow_add_entry": (_i, [_vp, _vp, _vp, _i, C.POINTER(_i)]),
    "bh_bow_erase_entry": (_i, [_vp, _i]),
    "bh_voc_create": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "bh_voc_destroy": (None, [_vp]),
    "bh_voc_transform": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i]),
    "bh_bow_query_l1": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i]),
    "bh_bow_query_l1_batch": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "bh_bow_query_l1_subset": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "bh_bow_compare_l1": (_i, [_vp, _i, _i, C.POINTER(_i), C.POINTER(_d)]),
}

_LIB = None
MISSING = []


def exported_symbols():
    """Names include/boslam_hip.h declares (kept in sync by tests/test_host_cpu.py)."""
    return sorted(_SIGS)


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Whichever copy is loaded first serves
    the whole process, and torch does not see the GPU through the system copy; so when a torch install exists and is not
    imported yet, load ITS runtime before libboslam_hip.so pulls in the system one.  torch itself is not imported."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("BH_NO_TORCH_HIP_PRELOAD"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    for root in (spec.submodule_search_locations or []) if spec else []:
        cand = os.path.join(root, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


_LIB_LOCK = threading.Lock()


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LIB_LOCK:                            # the tracker and the local-mapping thread may both make the first call
        if _LIB is None:
            if not os.path.exists(LIB_PATH):
                raise BhError(BH_ERR_HIP, f"{LIB_PATH} is not built (run __graft_entry__.build()); there is no CPU fallback")
            _preload_hip_runtime()
            L = C.CDLL(LIB_PATH)
            for name, (res, args) in _SIGS.items():
                fn = getattr(L, name, None)
                if fn is None:                     # tests/test_host_cpu.py asserts this never happens
                    MISSING.append(name)
                    continue
                fn.restype, fn.argtypes = res, args
            _LIB = L
    return _LIB


def ptr(a):
    """Pointer of a numpy array / int device address / None."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):             # torch tensor (device memory plumbing only)
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))


class Context:
    """One bh_ctx (own HIP stream + device workspace).  Use one per calling thread."""

    def __init__(self, device=0):
        h = C.c_void_p()
        rc = lib().bh_ctx_create(device, C.byref(h))
        if rc != BH_OK:
            raise BhError(rc, "bh_ctx_create failed: no usable HIP device (the HIP path is mandatory, no fallback)")
        self.h = h
        self.device = device

    def check(self, rc):
        if rc != BH_OK:
            raise BhError(rc, lib().bh_last_error(self.h).decode())

    def set_stream(self, stream_handle):
        self.check(lib().bh_ctx_set_stream(self.h, C.c_void_p(stream_handle) if stream_handle else None))

    def synchronize(self):
        self.check(lib().bh_ctx_synchronize(self.h))

    def status(self):
        self.check(lib().bh_ctx_status(self.h))

    def timing(self, enable):
        self.check(lib().bh_ctx_timing(self.h, 1 if enable else 0))

    def timing_read(self):
        """{kernel name: (total ms, launches)} since timing was enabled."""
        ms = (C.c_double * 32)(); cnt = (C.c_longlong * 32)(); names = (C.c_char_p * 32)(); n = C.c_int(0)
        self.check(lib().bh_ctx_timing_read(self.h, 32, ms, cnt, names, C.byref(n)))
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n.value) if cnt[i]}

    def close(self):
        if getattr(self, "h", None):
            lib().bh_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT = {}
_DEFAULT_LOCK = threading.Lock()


def default_context(device=0):
    """The calling thread's own context on `device` (the reference calls from two threads, slam/main.py:37-47: one bh_ctx each)."""
    key = (threading.get_ident(), device)
    with _DEFAULT_LOCK:
        c = _DEFAULT.get(key)
        if c is None:
            c = _DEFAULT[key] = Context(device)
        return c
