#!/bin/bash
# set_problem at configs[4] (warm: buffers exist), a kernel trace of one call.   usage (gpurun): bash tools/bld_time.sh <tag>
T=$1
python tools/ba_global.py > gpurun_out/bldt_$T.json 2> gpurun_out/bldt_$T.err
python - <<PY
import json
t=open("gpurun_out/bldt_$T.json").read()
d=json.loads(t[t.index("{"):])
print(d["value"], d["ms_per_iter"], "set_problem_s", d["set_problem_s"], "warm ms", d["set_problem_ms_warm"])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bldt; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bldt -- python3 $GRAFT_REPO_ROOT/tools/ba_global.py > /tmp/bldt.log 2>&1
f=$(find /tmp/bldt -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/bldt_${T}_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/bldt_${T}_kernel_stats.csv")))
for r in rows:
    n=r['Name']
    if n.startswith('k_bld') or n.startswith('k_g_') or 'scan' in n or 'k_ba_schur_r' in n or 'k_ba_schur_aux' in n or 'rocprim' in n or 'Buffer' in n:
        print(f"{n[:50]:50s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us tot {float(r['TotalDurationNs'])/1e6:7.2f} ms")
PY
