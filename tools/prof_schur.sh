#!/bin/bash
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/tools/ba_global.py > $OUT/a.log 2>&1 || echo fail-a
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 $GRAFT_REPO_ROOT/tools/ba_global.py > $OUT/f.log 2>&1 || echo fail-f
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/tools/ba_global.py > $OUT/b.log 2>&1 || echo fail-b
echo done
