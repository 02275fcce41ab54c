import sys, os, json; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from boslam_amd import Context, bow_bench
ctx = Context(0)
print(json.dumps(bow_bench.run_sharded(ctx, repeats=10)))
