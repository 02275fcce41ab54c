"""bh_ba_set_problem timing: device-built structure vs the former host loops (BH_BA_BUILD=host), configs[2] and configs[4]."""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from boslam_amd import Context, ba, synth
ctx = Context(0)
for name, pr in (("config3", synth.ba_problem()), ("config5", synth.ba_problem_sequence())):
    for mode in ("device", "host"):
        if mode == "host": os.environ["BH_BA_BUILD"] = "host"
        else: os.environ.pop("BH_BA_BUILD", None)
        s = ba.LocalBA(ctx, pr["K"])
        ts = []
        for r in range(4):
            t0 = time.perf_counter(); s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"]); ts.append(time.perf_counter() - t0)
        print(f"{name} {mode}: set_problem {1e3 * min(ts[1:]):.2f} ms (first call {1e3 * ts[0]:.1f} ms), {len(pr['ecam'])} edges")
        s.close()
