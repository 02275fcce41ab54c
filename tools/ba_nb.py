import sys, os, json
sys.path.insert(0, os.getcwd())
from boslam_amd import Context, ba_bench
ctx = Context(0)
r = ba_bench.run(ctx)
print(os.environ.get("BH_BA_CHOL_NB"), r["value"], r["ms_per_iter"], r["kernel_ms_per_iter"])
