#!/bin/bash
# refresh of the three BA kernel-stats profiles only (GPU box): bash tools/prof_ba_only.sh <outdir-under-gpurun_out>
KEEP=$GRAFT_REPO_ROOT/gpurun_out/$1; OUT=/tmp/prof_$1; rm -rf $OUT; mkdir -p $OUT $KEEP
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba_local -- python3 $GRAFT_REPO_ROOT/tools/ba_only.py > $OUT/ba_local.log 2>&1 || echo "ba_local failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba_global -- python3 $GRAFT_REPO_ROOT/tools/ba_global.py > $OUT/ba_global.log 2>&1 || echo "ba_global failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba_batch -- python3 $GRAFT_REPO_ROOT/tools/ba_batch.py 64 > $OUT/ba_batch.log 2>&1 || echo "ba_batch failed"
for d in ba_local ba_global ba_batch; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $KEEP/${d}_kernel_stats.csv; done
echo done
