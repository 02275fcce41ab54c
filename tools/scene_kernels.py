"""Per-kernel front-end times for the three synthetic scenes of bench.py (default, low_texture, high_corner_density), 1 000 frames each:
frames/s with the library's default pipelines and ms per step per kernel from a one-pipeline pass (ctx.timing).   usage (gpurun): python tools/scene_kernels.py"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from boslam_amd import Context, frontend

dev = torch.device("cuda:0")
ctx = Context(0)
params = frontend.make_params(nfeatures=1000)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
out = {}
for scene in ("default", "low_texture", "high_corner_density"):
    sb, sd = bench.gen_frames_device(torch, B, dev, scene=scene)
    f = frontend.DeviceFrontEnd(ctx, B, 480, 640, params, torch_device=dev)
    for _ in range(2): f.run(sb, sd)
    torch.cuda.synchronize(); ctx.status()
    t0 = time.perf_counter()
    for _ in range(5): f.run(sb, sd)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    os.environ["BH_ORB_PIPELINES"] = "1"
    f.run(sb, sd); torch.cuda.synchronize()
    ctx.timing(True)
    for _ in range(3): f.run(sb, sd)
    kt = ctx.timing_read(); ctx.timing(False)
    os.environ.pop("BH_ORB_PIPELINES", None)
    out[scene] = {"frames_per_s": round(B * 5 / el, 1), "mean_kp": float(f.n_kp.float().mean().item()),
                  "kernel_ms_per_step": {k: round(v[0] / 3, 3) for k, v in kt.items() if v[1] > 0}}
    del f, sb, sd
print(json.dumps(out, indent=1))
ctx.close()
