import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from boslam_amd import Context, track_bench
ctx = Context(0)
r = track_bench.run(ctx)
print(json.dumps({k: r[k] for k in ("pose_only_ba", "pnp_ransac")}))
