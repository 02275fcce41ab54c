"""How much of k_pyr_down / k_fast is the mask plane?  The same batch with and without a depth mask, per-kernel hipEvent times."""
import sys, os, ctypes as C; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ["BH_ORB_PIPELINES"] = "1"
import torch
from boslam_amd import Context, frontend, _lib
import bench
ctx = Context(0)
B = 512
bgr, depth = bench.gen_frames_device(torch, B, "cuda:0")
params = frontend.make_params(nfeatures=1000)
fe = frontend.DeviceFrontEnd(ctx, B, 480, 640, params)
L = _lib.lib()
for name, dptr in (("with mask", _lib.ptr(depth)), ("no mask", None)):
    def run():
        ctx.check(L.bh_orb_detect_describe(ctx.h, _lib.ptr(bgr), 3, None, dptr, B, 480, 640, C.byref(params), fe.max_kp, _lib.ptr(fe.kps), _lib.ptr(fe.desc), _lib.ptr(fe.n_kp), 1))
    run(); ctx.synchronize()
    ctx.timing(True)
    for _ in range(4): run()
    kt = ctx.timing_read(); ctx.timing(False)
    print(name, {k: round(v[0] / 4 / B * 1000, 3) for k, v in sorted(kt.items())}, "ms per 1000 frames")
