#!/bin/bash
# timeline (kernels + copies) of the LAST bh_ba_set_problem of tools/ba_only.py on the GPU box: bash tools/trace_setup2.sh
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_sp -o t -- python3 $GRAFT_REPO_ROOT/tools/ba_only.py > /tmp/sp.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("/tmp/prof_sp/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
for f in glob.glob("/tmp/prof_sp/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_b2_mark")]
s = idx[-1]
# walk back to the first copy of this set_problem (gap > 300 us = previous call)
a = s
while a > 0 and rows[a][0] - rows[a - 1][1] < 300000: a -= 1
b = s
while b + 1 < len(rows) and not rows[b + 1][2].startswith("k_bb_linearize") and not rows[b + 1][2].startswith("k_ba_linearize"): b += 1
t0 = rows[a][0]; prev = None
for st, en, n in rows[a:b + 2]:
    print(f"{(st - t0) / 1e3:9.1f} us  +{((st - prev) / 1e3 if prev else 0):6.1f} gap  {(en - st) / 1e3:7.1f} us  {n[:50]}")
    prev = en
PY
