"""64 batched windows + the single window, live size and configs[2]: python tools/ba_batch_ab.py  (A/B switches through the environment)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from boslam_amd import Context, ba_bench
ctx = Context(0)
for kw in (dict(n_kf=10, n_pts=600, iters=7, label="live"), dict()):
    d = ba_bench.run_batch(ctx, W=64, **kw)
    print({k: d.get(k) for k in ("value", "ms_per_round", "single_window_iters_per_s", "speedup_vs_one_at_a_time", "value_incl_set_problem", "single_window_iters_per_s_incl_set_problem")}, flush=True)
