"""Phase clocks of k_pnp_ransac on the bench workload - needs a side library built from a pnp.hip that writes its clock deltas into
rvec / tvec (see the round-2 notes in DESIGN.md section 5); prints mean microseconds per phase and hypotheses per problem."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from boslam_amd import Context, pnp, synth
ctx = Context(0)
rng = np.random.default_rng(1)
B, S = 1000, 250
K4 = np.array([synth.FX, synth.FY, synth.CX, synth.CY])
obj = rng.uniform(-1, 1, (B, S, 3)) + np.array([0, 0, 4.0])
img = np.stack([synth.FX * obj[..., 0] / obj[..., 2] + synth.CX, synth.FY * obj[..., 1] / obj[..., 2] + synth.CY], -1)
img += rng.normal(0, 0.5, img.shape)
n = np.full(B, S, np.int32)
for _ in range(2):
    rv, tv, inl, ni, ok = pnp.solve_pnp_ransac_batch(ctx, obj, img, n, K4)
print("rng %.1f us, ransac %.1f us, mask %.1f us, LM + output %.1f us; hypotheses per problem: mean %.2f max %d" %
      (rv[:, 0].mean() / 100, rv[:, 1].mean() / 100, rv[:, 2].mean() / 100, tv[:, 0].mean() / 100, tv[:, 1].mean(), tv[:, 1].max()))
print("slowest problem: total %.1f us" % ((rv[:, 0] + rv[:, 1] + rv[:, 2] + tv[:, 0]).max() / 100))
