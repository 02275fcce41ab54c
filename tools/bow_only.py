"""configs[3] BoW query on one GPU (both paths): python tools/bow_only.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boslam_amd import Context, bow_bench
ctx = Context(0)
print(json.dumps(bow_bench.run_sharded(ctx, None, "cuda:0")))
