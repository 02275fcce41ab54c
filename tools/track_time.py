import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from boslam_amd import Context, ba, pnp, synth
ctx = Context(0)
K4 = np.array([synth.FX, synth.FY, synth.CX, synth.CY])
for B, S in ((100, 250), (1000, 250), (1000, 600)):
    rng = np.random.default_rng(1)
    obj = rng.uniform(-1, 1, (B, S, 3)) + np.array([0, 0, 4.0])
    img = np.stack([synth.FX * obj[..., 0] / obj[..., 2] + synth.CX, synth.FY * obj[..., 1] / obj[..., 2] + synth.CY], -1) + rng.normal(0, 0.5, (B, S, 2))
    n = np.full(B, S, np.int32)
    pnp.solve_pnp_ransac_batch(ctx, obj, img, n, K4)
    t0 = time.perf_counter(); r = pnp.solve_pnp_ransac_batch(ctx, obj, img, n, K4); dt = time.perf_counter() - t0
    print(f"pnp_ransac  B={B} S={S}: {dt*1e3:.2f} ms  ({B/dt:.0f} problems/s, incl. host staging)  ok={r[4].mean():.2f}")
    K5 = np.array([synth.FX, synth.FY, synth.CX, synth.CY, 0.0])
    ba.pose_only_ba_batch(ctx, img, obj, n, K5)
    t0 = time.perf_counter(); r = ba.pose_only_ba_batch(ctx, img, obj, n, K5); dt = time.perf_counter() - t0
    print(f"pose_only_ba B={B} S={S}: {dt*1e3:.2f} ms  ({B/dt:.0f} problems/s, incl. host staging)")
ctx.timing(True)
pnp.solve_pnp_ransac_batch(ctx, obj, img, n, K4); ba.pose_only_ba_batch(ctx, img, obj, n, K5)
print(ctx.timing_read())
