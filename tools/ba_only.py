import sys, json; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from boslam_amd import Context, ba_bench
ctx = Context(0); print(json.dumps(ba_bench.run(ctx), indent=1)); ctx.close()
