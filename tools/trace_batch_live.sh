#!/bin/bash
# kernel timeline of the batched local BA at the live window size (GPU box): bash tools/trace_batch_live.sh [W]
cd /tmp && export TMPDIR=/tmp
cat > /tmp/bl.py <<PY
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from boslam_amd import Context, ba_bench
ctx = Context(0)
print(json.dumps(ba_bench.run_batch(ctx, W=int(sys.argv[1]), n_kf=10, n_pts=600, iters=7, label="live")))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bl -o bl -- python3 /tmp/bl.py ${1:-64} > /tmp/bl.log 2>&1
grep value /tmp/bl.log | cut -c1-400
python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/prof_bl/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:22]:
        print(r["Name"][:70].ljust(70), r["Calls"], r["AverageNs"])
PY
