#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench command with one chunk pipeline (the configuration bench.py's roofline pass
# times: with several pipelines in flight a launch shares the chip and single-launch durations are not additive).
# Usage on the GPU box: bash tools/prof_stats.sh <outdir-under-gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 --pipelines 1 > $OUT/stats.log 2>&1
echo stats-done
