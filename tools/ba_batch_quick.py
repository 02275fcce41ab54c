"""64 batched windows, live size and configs[2]: python tools/ba_batch_quick.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from boslam_amd import Context, ba_bench
ctx = Context(0)
for kw in (dict(n_kf=10, n_pts=600, iters=7, label="live"), dict()):
    d = ba_bench.run_batch(ctx, W=64, **kw)
    print({k: d[k] for k in ("value", "ms_per_round", "speedup_vs_one_at_a_time", "value_incl_set_problem", "speedup_incl_set_problem")}, flush=True)
