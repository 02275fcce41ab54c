"""A/B of the local-BA bench: prints value / ms per iteration / kernel times / batched figures for each JSON produced by tools/ba_only.py."""
import json, sys
for f in sys.argv[1:]:
    t = open(f).read(); d = json.loads(t[t.index("{"):])
    b = d.get("batched", {})
    print(f, d["value"], d["ms_per_iter"], d.get("kernel_ms_per_iter"), "live", b.get("live_windows", {}).get("value"),
          b.get("live_windows", {}).get("single_window_iters_per_s"), "cfg2", b.get("configs2_windows", {}).get("value"))
