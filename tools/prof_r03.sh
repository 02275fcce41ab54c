#!/bin/bash
# Round-3 profile set (run on the GPU box from the repo root: bash tools/prof_r03.sh <outdir-under-gpurun_out>):
#   stats      rocprofv3 --kernel-trace --stats of the bench command with ONE chunk pipeline (what bench.py's roofline pass times)
#   sq1 sq2 fetch write grbm   PMC passes of the front end, one counter group per pass (gfx950 slot limits; --pmc never with the
#              hip/hsa trace domains)
#   bow_fetch bow_write        FETCH_SIZE / WRITE_SIZE of the configs[3] BoW query (both paths)
#   ba_local ba_global ba_batch  kernel stats of the local, the global and the batched local BA
set -e
KEEP=$GRAFT_REPO_ROOT/gpurun_out/$1
OUT=/tmp/prof_$1          # raw rocprofv3 output stays on the box (hundreds of MB); only the summaries are copied to gpurun_out
rm -rf $OUT; mkdir -p $OUT $KEEP
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py"
FE="--frames 256 --steps 2 --warmup 1 --no-ba --no-global-ba --no-bow --no-inputs --cpu-sample 0 --pipelines 1"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 5 --warmup 2 --cpu-sample 0 --pipelines 1 --no-inputs --no-ba --no-global-ba --no-bow > $OUT/stats.log 2>&1 || echo "stats failed"
echo stats-done
run() { local name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $B $FE > $OUT/$name.log 2>&1 || echo "pass $name failed"; echo $name-done; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/bow_fetch -- python3 $GRAFT_REPO_ROOT/tools/bow_only.py > $OUT/bow_fetch.log 2>&1 || echo "bow_fetch failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/bow_write -- python3 $GRAFT_REPO_ROOT/tools/bow_only.py > $OUT/bow_write.log 2>&1 || echo "bow_write failed"
echo bow-done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba_local -- python3 $GRAFT_REPO_ROOT/tools/ba_only.py > $OUT/ba_local.log 2>&1 || echo "ba_local failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba_global -- python3 $GRAFT_REPO_ROOT/tools/ba_global.py > $OUT/ba_global.log 2>&1 || echo "ba_global failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba_batch -- python3 $GRAFT_REPO_ROOT/tools/ba_batch.py 64 > $OUT/ba_batch.log 2>&1 || echo "ba_batch failed"
# summaries -> gpurun_out
cd $GRAFT_REPO_ROOT
for d in stats ba_local ba_global ba_batch; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $KEEP/${d}_kernel_stats.csv; done
python3 tools/pmc_summary.py $OUT > $KEEP/pmc_summary.json 2> $KEEP/pmc_summary.err || echo "pmc_summary failed"
python3 tools/pmc_traffic.py $OUT 128 > $KEEP/pmc_traffic.json 2> $KEEP/pmc_traffic.err || echo "pmc_traffic failed"
for l in stats sq1 sq2 fetch write grbm bow_fetch bow_write ba_local ba_global ba_batch; do tail -c 3000 $OUT/$l.log > $KEEP/$l.log.tail 2>/dev/null || true; done
du -sh $OUT | tail -1
echo prof-done
