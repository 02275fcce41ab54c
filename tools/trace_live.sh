#!/bin/bash
# kernel timeline of the one-frame-per-call tracking loop (frontend.LiveTracker) on the GPU box: bash tools/trace_live.sh
cd /tmp && export TMPDIR=/tmp
cat > /tmp/lv.py <<PY
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from boslam_amd import Context, frontend, synth
ctx = Context(0)
bgr, depth = synth.render_batch(2)
params = frontend.make_params(nfeatures=1000)
K = np.array([synth.FX, synth.FY, synth.CX, synth.CY, 0.06])
lt = frontend.LiveTracker(ctx, 480, 640, params, K)
lat = []
for i in range(40):
    t0 = time.perf_counter(); o = lt.step(bgr[i & 1], depth[i & 1]); ctx.synchronize(); lat.append(time.perf_counter() - t0)
print("median ms", 1e3 * float(np.median(lat[5:])), o["status"], o["n_inlier"])
lt.close()
PY
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_lv -o lv -- python3 /tmp/lv.py > /tmp/lv.log 2>&1
grep median /tmp/lv.log
python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/prof_lv/**/*kernel_stats.csv", recursive=True):
    tot = 0
    for r in list(csv.DictReader(open(f))):
        if int(r["Calls"]) >= 39:
            print(r["Name"][:60].ljust(60), r["Calls"], r["AverageNs"]); tot += float(r["AverageNs"]) * int(r["Calls"]) / 40
    print("kernel us per frame", tot / 1e3)
for f in glob.glob("/tmp/prof_lv/**/*memory_copy_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"], r["Calls"], r["AverageNs"])
PY
