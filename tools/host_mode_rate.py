"""PCIe-inclusive rate of the headline path: the same ORB detect + describe + consecutive-frame cross-check match as bench.py's step, but through the
BH_MEM_HOST entry points - frames and depth in host (pageable numpy) memory, keypoints / descriptors / matches back in host memory, every call synchronous
(staged through the context's workspace in chunks of 256 frames).  Never bench.py's `value` (inputs resident in HBM); DESIGN.md section 6 quotes it beside it.
Also times the C entry point alone with pre-touched output arrays (the mirror allocates fresh, zeroed outputs per call).
usage (gpurun): python tools/host_mode_rate.py [frames]"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import numpy as np
import torch
import bench
from boslam_amd import Context, frontend, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
sb, sd = bench.gen_frames_device(torch, B, dev)
bgr = sb.cpu().numpy(); depth = sd.cpu().numpy().view(np.uint16)      # (the generator's int16 tensor: the same bits; as int16 the mirror would convert 614 MB per call)
del sb, sd
ctx = Context(0)
params = frontend.make_params(nfeatures=1000)
out = {}
for rep in range(3):
    t0 = time.perf_counter()
    kps, desc, n = frontend.orb_detect_describe(ctx, bgr, depth=depth, params=params)
    t1 = time.perf_counter()
    qi, ti, d, nm = frontend.match_pairs(ctx, desc[1:], n[1:], desc[:-1], n[:-1])
    t2 = time.perf_counter()
    out[rep] = {"orb_ms": round(1e3 * (t1 - t0), 2), "match_ms": round(1e3 * (t2 - t1), 2), "frames_per_s": round(B / (t2 - t0), 1)}
bytes_in = bgr.nbytes + depth.nbytes
best = max(out.values(), key=lambda r: r["frames_per_s"])
# the C entry point alone, outputs allocated and touched beforehand
max_kp = kps.shape[1]
H, W = bgr.shape[1:3]
kps2 = np.ones((B, max_kp), frontend.KP_DTYPE); desc2 = np.ones((B, max_kp, 32), np.uint8); n2 = np.ones(B, np.int32)
H, W = bgr.shape[1:3]
order = []
for rep in range(4):                                               # mirror and C entry alternating: is the difference the call path or the call count?
    t0 = time.perf_counter(); frontend.orb_detect_describe(ctx, bgr, depth=depth, params=params); order.append(("mirror", round(1e3 * (time.perf_counter() - t0), 2)))
    t0 = time.perf_counter()
    _lib.lib().bh_orb_detect_describe(ctx.h, _lib.ptr(bgr), 3, None, _lib.ptr(depth), B, H, W, C.byref(params), max_kp, _lib.ptr(kps2), _lib.ptr(desc2), _lib.ptr(n2), 0)
    order.append(("c_reused_outputs", round(1e3 * (time.perf_counter() - t0), 2)))
    k3 = np.ones((B, max_kp), frontend.KP_DTYPE); d3 = np.ones((B, max_kp, 32), np.uint8); n3 = np.ones(B, np.int32)
    t0 = time.perf_counter()
    _lib.lib().bh_orb_detect_describe(ctx.h, _lib.ptr(bgr), 3, None, _lib.ptr(depth), B, H, W, C.byref(params), max_kp, _lib.ptr(k3), _lib.ptr(d3), _lib.ptr(n3), 0)
    order.append(("c_fresh_outputs", round(1e3 * (time.perf_counter() - t0), 2)))
c_ms = []
for rep in range(3):
    t0 = time.perf_counter()
    rc = _lib.lib().bh_orb_detect_describe(ctx.h, _lib.ptr(bgr), 3, None, _lib.ptr(depth), B, H, W, C.byref(params), max_kp, _lib.ptr(kps2), _lib.ptr(desc2), _lib.ptr(n2), 0)
    c_ms.append(round(1e3 * (time.perf_counter() - t0), 2))
    assert rc == 0
assert np.array_equal(n2, n) and np.array_equal(desc2[0, :n[0]], desc[0, :n[0]])
t0 = time.perf_counter(); z = np.zeros((B, max_kp, 32), np.uint8); z2 = np.zeros((B, max_kp), frontend.KP_DTYPE); z[:] = 1; z2["x"] = 1; alloc_ms = round(1e3 * (time.perf_counter() - t0), 2)
print(json.dumps({"frames": B, "mean_kp": float(n.mean()), "mean_matches": float(nm.mean()), "runs": out, "best": best,
                  "c_entry_ms_pretouched_outputs": c_ms, "alternating": order, "alloc_and_touch_outputs_ms": alloc_ms, "max_kp": int(max_kp),
                  "host_bytes_in_per_frame": bytes_in // B, "h2d_GBps_implied_by_orb_call": round(bytes_in / (best["orb_ms"] * 1e-3) / 1e9, 2)}, indent=1))
ctx.close()
