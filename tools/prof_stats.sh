#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command (same flags the driver uses, fewer steps).
# Usage on the GPU box: bash tools/prof_stats.sh <outdir-under-gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $OUT/stats.log 2>&1
echo stats-done
