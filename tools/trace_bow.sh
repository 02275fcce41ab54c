#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_bow
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bow_only.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v); print("%-28s n %5d  med %8.1f us  sum %9.0f" % (k[:28], len(v), v[len(v)//2], sum(v)))
# one inverted query: sequence with gaps
idx = [i for i, r in enumerate(rows) if "k_inv_gather" in r["Kernel_Name"]]
i = idx[len(idx)//2]
t0 = int(rows[i]["Start_Timestamp"])
for r in rows[i-2:i+6]:
    print("%-24s start %8.1f dur %7.1f" % (r["Kernel_Name"].split("(")[0][:24], (int(r["Start_Timestamp"]) - t0) / 1000, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000))
PY
