#!/bin/bash
# kernel + copy timeline of bh_ba_set_problem at the live window size (GPU box): bash tools/trace_setup.sh [modes] [cases]
cd /tmp && export TMPDIR=/tmp
BA_SETUP_CASES=${2:-0} BA_SETUP_MODES=${1:-small} rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_bs -o bs -- python3 $GRAFT_REPO_ROOT/tools/ba_setup.py > /tmp/bs.log 2>&1
grep set_problem /tmp/bs.log
python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/prof_bs/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:16]:
        print(r["Name"][:70].ljust(70), r["Calls"], r["AverageNs"])
for f in glob.glob("/tmp/prof_bs/**/*memory_copy_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"], r["Calls"], r["AverageNs"])
PY
