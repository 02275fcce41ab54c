"""Batched local BA timing (bh_ba_batch_optimize): python tools/ba_batch.py [W]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boslam_amd import Context, ba_bench
ctx = Context(0)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 64
print(json.dumps(ba_bench.run_batch(ctx, W=W)))
print(json.dumps(ba_bench.run_batch(ctx, W=W, n_kf=10, n_pts=600, iters=7, label="live size (covisibility_graph.py:220: 7 iterations)")))
print(json.dumps(ba_bench.run(ctx)))
