import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from boslam_amd import Context, ba, synth
ctx = Context(0)
pr = synth.ba_problem()
s = ba.LocalBA(ctx, pr["K"])
for r in range(5):
    t0 = time.perf_counter(); s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"]); t1 = time.perf_counter()
    it = s.optimize(7); ctx.synchronize(); t2 = time.perf_counter()
    R, t = s.poses(); p = s.points(); c = s.chi2(); t3 = time.perf_counter()
    print(f"set_problem {1e3*(t1-t0):.2f} ms  optimize(7) {1e3*(t2-t1):.2f} ms  read-back {1e3*(t3-t2):.2f} ms")
