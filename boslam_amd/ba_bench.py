"""Secondary metric of BASELINE.json: local-BA Levenberg iterations per second on configs[2]
(50 keyframes x 10 k landmarks x ~80 k observations, 10 LM iterations, Huber sqrt(7.815), f64)."""
import time
import numpy as np
from . import ba, synth


def run(ctx, dev=None, repeats=5):
    pr = synth.ba_problem()
    s = ba.LocalBA(ctx, pr["K"])
    times, iters = [], 0
    for r in range(repeats + 1):
        s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
        ctx.synchronize()
        t0 = time.perf_counter()
        iters = s.optimize(10)
        ctx.synchronize()
        if r:
            times.append(time.perf_counter() - t0)
    ctx.timing(True)                                   # one more pass with per-kernel hipEvent timing
    s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    s.optimize(10)
    kt = ctx.timing_read()
    ctx.timing(False)
    lg = s.log()[:iters]
    _, t = s.poses()
    out = {"metric": "local_ba_lm_iters_per_sec", "value": round(iters / float(np.median(times)), 1), "unit": "iters/s",
           "iters": int(iters), "ms_per_iter": round(float(np.median(times)) / iters * 1e3, 3), "dtype": "f64",
           "workload": f"configs[2]: {len(pr['R'])} keyframes ({int(pr['fixed'].sum())} fixed) x {len(pr['pts'])} landmarks x "
                       f"{len(pr['ecam'])} observations, 10 LM iterations",
           "final_robust_chi2": float(lg[-1, 0]), "trials": int(lg[:, 2].sum()),
           "kernel_ms_per_iter": {k: round(v[0] / iters, 4) for k, v in sorted(kt.items()) if k.startswith("k_ba")},
           "pose_err_before_m": float(np.abs(pr["t"] - pr["tgt"]).max()), "pose_err_after_m": float(np.abs(t - pr["tgt"]).max())}
    s.close()
    return out
