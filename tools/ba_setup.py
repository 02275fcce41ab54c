"""bh_ba_set_problem wall time by builder (one-workgroup / chip-wide / host loops) at the live window size and at configs[2].
usage (GPU box): python tools/ba_setup.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from boslam_amd import Context, ba, synth

ctx = Context(0)
CASES = [int(v) for v in os.environ.get("BA_SETUP_CASES", "0,1,2").split(",")]
MODES = os.environ.get("BA_SETUP_MODES", "small,general,host").split(",")
for ci, (name, kw) in enumerate((("live 10x600", dict(n_kf=10, n_pts=600, seed=1)), ("20x2000", dict(n_kf=20, n_pts=2000, seed=2)), ("configs[2] 50x10000", dict(n_kf=50, n_pts=10000, seed=7)))):
    if ci not in CASES:
        continue
    pr = synth.ba_problem(**kw)
    for mode in MODES:
        if mode == "small":
            os.environ.pop("BH_BA_BUILD", None)
        else:
            os.environ["BH_BA_BUILD"] = mode
        s = ba.LocalBA(ctx, pr["K"])
        for _ in range(3):
            s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
        ctx.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / n
        t1 = time.perf_counter()
        s.optimize(7)
        ctx.synchronize()
        print(f"{name:22s} {mode:8s} set_problem {dt * 1e3:7.3f} ms   (7 iterations {1e3 * (time.perf_counter() - t1):.3f} ms, E = {len(pr['ecam'])})", flush=True)
        s.close()
os.environ.pop("BH_BA_BUILD", None)
# many windows in one call
for W in (16, 64):
    prs = [synth.ba_problem(n_kf=10, n_pts=600, seed=100 + i) for i in range(W)]
    wins = [ba.LocalBA(ctx, pr["K"]) for pr in prs]
    args = [(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"]) for pr in prs]
    for _ in range(2):
        ba.set_problems(ctx, wins, args)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ba.set_problems(ctx, wins, args)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(3):
        for w, a in zip(wins, args):
            w.set_problem(*a)
    ctx.synchronize()
    d1 = (time.perf_counter() - t0) / 3
    b = ba.LocalBABatch(ctx, wins)
    b.optimize(7)
    ba.set_problems(ctx, wins, args)
    ctx.synchronize()
    t0 = time.perf_counter()
    b.optimize(7)
    ctx.synchronize()
    do = time.perf_counter() - t0
    print(f"{W} live windows: set_problems {dt * 1e3:.3f} ms ({dt / W * 1e3:.4f} per window; one by one {d1 / W * 1e3:.4f}); batched optimize(7) {do * 1e3:.3f} ms", flush=True)
    b.close()
    for w in wins:
        w.close()
