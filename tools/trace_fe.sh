#!/bin/bash
# kernel trace of the front end (one pipeline), per-kernel/per-grid median durations.  usage: bash tools/trace_fe.sh <tag> [env assignments...]
T=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$T
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --frames 512 --steps 3 --warmup 1 --no-ba --no-global-ba --no-bow --no-inputs --cpu-sample 0 --pipelines 1 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d[(k, r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if k[0].startswith("k_"):
        v = sorted(v); print("%-22s grid %7s x %4s  n %4d  med %7.1f us  sum %8.0f" % (k[0][:22], k[1], k[2], len(v), v[len(v)//2], sum(v)))
PY
