import sys, json; sys.path.insert(0, '.')
from boslam_amd import Context, ba_bench
ctx = Context(0); print(json.dumps(ba_bench.run_global(ctx), indent=1)); ctx.close()
