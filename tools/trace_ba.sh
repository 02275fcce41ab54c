#!/bin/bash
# kernel timeline of the local-BA bench (tools/ba_only.py) on the GPU box: bash tools/trace_ba.sh <tag> [env ...]; prints the kernels of the LAST
# optimize call with start offsets and gaps (us)
T=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ba_$T
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ba_$T -o t -- python3 $GRAFT_REPO_ROOT/tools/ba_only.py > /tmp/ba_$T.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("/tmp/prof_ba_$T/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
# the last run of 10 iterations of the single window: find the last k_ba_linearize x10 block
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_ba_linearize")]
start = idx[-10] if len(idx) >= 10 else 0
end = idx[-1]
# extend to the end of that iteration
j = end
while j + 1 < len(rows) and not rows[j + 1][2].startswith("k_b2") and not rows[j + 1][2].startswith("k_bb") and rows[j + 1][0] - rows[j][1] < 200000: j += 1
t0 = rows[start][0]; prev_end = None
tot_gap = 0; tot_k = 0
for s, e, n in rows[start:j + 1]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    if n.startswith("k_ba_linearize"): print("----")
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:6.1f} gap  {(e - s) / 1e3:7.1f} us  {n[:40]}")
    tot_gap += gap; tot_k += (e - s) / 1e3; prev_end = e
print("span", (rows[j][1] - t0) / 1e3, "kernels", tot_k, "gaps", tot_gap)
PY
