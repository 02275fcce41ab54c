#!/bin/bash
# Collect rocprofv3 PMC counters for the bench kernels, one counter group per pass (gfx950 slot limits:
# SQ 8, TCC 4 with FETCH_SIZE=3 / WRITE_SIZE=2).  Usage (on the GPU box, from the repo root):
#   bash tools/prof_pmc.sh <outdir-under-gpurun_out> [bench args...]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
ARGS=${@:---frames 256 --steps 2 --warmup 1 --no-ba --no-global-ba --no-bow --no-inputs --cpu-sample 0 --pipelines 1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
echo done
