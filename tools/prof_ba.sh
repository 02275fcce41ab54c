#!/bin/bash
# rocprofv3 on the BA paths: kernel stats of local BA (configs[2]) + global BA (configs[4]) and the matrix-core counters of the
# reduced-camera-system solver.  Usage on the GPU box: bash tools/prof_ba.sh <outdir-under-gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counters.txt || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_global -- python3 $GRAFT_REPO_ROOT/tools/ba_global.py > $OUT/stats_global.log 2>&1 || echo "stats_global failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_local -- python3 $GRAFT_REPO_ROOT/tools/ba_only.py > $OUT/stats_local.log 2>&1 || echo "stats_local failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_global -- python3 $GRAFT_REPO_ROOT/tools/ba_global.py > $OUT/pmc_global.log 2>&1 || echo "pmc_global failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_local -- python3 $GRAFT_REPO_ROOT/tools/ba_only.py > $OUT/pmc_local.log 2>&1 || echo "pmc_local failed"
echo ba-prof-done
