"""Tracking-stage throughput: batched pose-only BA, solvePnPRansac and the visibility gate on
synthetic problems of the reference's live size (250 matched points per frame).  Host arrays in, host arrays out: the figures
include the PCIe staging of the call (small: ~10 KB per problem)."""
import time
import numpy as np
from . import align, ba, pnp, synth


def run(ctx, B=1000, S=250):
    rng = np.random.default_rng(1)
    K4 = np.array([synth.FX, synth.FY, synth.CX, synth.CY]); K5 = np.array([synth.FX, synth.FY, synth.CX, synth.CY, 0.0])
    obj = rng.uniform(-1, 1, (B, S, 3)) + np.array([0, 0, 4.0])
    img = np.stack([synth.FX * obj[..., 0] / obj[..., 2] + synth.CX, synth.FY * obj[..., 1] / obj[..., 2] + synth.CY], -1)
    img += rng.normal(0, 0.5, img.shape)
    n = np.full(B, S, np.int32)
    out = {"workload": f"{B} independent problems x {S} points (0.5 px noise)"}
    for name, fn in (("pose_only_ba", lambda: ba.pose_only_ba_batch(ctx, img, obj, n, K5)),
                     ("pnp_ransac", lambda: pnp.solve_pnp_ransac_batch(ctx, obj, img, n, K4))):
        fn()
        ctx.timing(True)
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
        kt = ctx.timing_read(); ctx.timing(False)
        out[name] = {"problems_per_s": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
                     "kernel_ms": {k: round(v[0], 3) for k, v in kt.items()}}
    # loop-closure verification: the same batch as 3-D / 3-D alignment problems (30 % of the pairs gross outliers)
    R0 = synth._rodrigues(np.array([0.1, -0.2, 0.15])); t0 = np.array([0.3, -0.1, 0.2])
    Q = obj @ R0.T + t0 + rng.normal(0, 0.004, obj.shape)
    bad = rng.random((B, S)) < 0.3
    Q[bad] += rng.normal(0, 1.0, (int(bad.sum()), 3))
    fn = lambda: align.align_se3_ransac_batch(ctx, obj, Q, n)
    fn()
    ctx.timing(True)
    t0_ = time.perf_counter(); o = fn(); dt = time.perf_counter() - t0_
    kt = ctx.timing_read(); ctx.timing(False)
    out["se3_alignment"] = {"problems_per_s": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "kernel_ms": {k: round(v[0], 3) for k, v in kt.items()},
                            "ok": int((o["status"] == 0).sum()), "mean_inliers": float(o["n_inlier"].mean()), "max_rot_err": float(np.abs(o["R"] - R0).max())}
    return out
