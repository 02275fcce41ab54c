import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from boslam_amd import Context, pnp, synth
ctx = Context(0)
K4 = np.array([synth.FX, synth.FY, synth.CX, synth.CY])
B, S = 64, 250
rng = np.random.default_rng(1)
obj = rng.uniform(-1, 1, (B, S, 3)) + np.array([0, 0, 4.0])
img = np.stack([synth.FX * obj[..., 0] / obj[..., 2] + synth.CX, synth.FY * obj[..., 1] / obj[..., 2] + synth.CY], -1) + rng.normal(0, 0.5, (B, S, 2))
n = np.full(B, S, np.int32)
for iters in (1, 2, 4, 100):
    pnp.solve_pnp_ransac_batch(ctx, obj, img, n, K4, iterations=iters)
    ctx.timing(True); pnp.solve_pnp_ransac_batch(ctx, obj, img, n, K4, iterations=iters); kt = ctx.timing_read(); ctx.timing(False)
    print("iterations", iters, "kernel ms", round(kt["k_pnp_ransac"][0], 3))
n5 = np.full(B, 5, np.int32)
ctx.timing(True); pnp.solve_pnp_ransac_batch(ctx, obj, img, n5, K4); kt = ctx.timing_read(); ctx.timing(False)
print("n=5 (one hypothesis, 5-point LM)", round(kt["k_pnp_ransac"][0], 3))
