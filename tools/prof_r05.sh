#!/bin/bash
# Round-5 profile set (same passes as round 4).  Run on the GPU box from the repo root:  bash tools/prof_r05.sh <outdir-under-gpurun_out> [pass ...]
# Passes (default: all).  One counter group per pass (gfx950 slot limits); --pmc never together with the hip / hsa trace domains; the
# program goes directly after "--".
#   stats                      rocprofv3 --kernel-trace --stats of the front-end bench with ONE chunk pipeline (what bench.py's roofline pass times)
#   sq1 sq2 fetch write grbm   PMC passes of the front end;  ta1 tcp1 tcc1: texture-address / L1 / L2 counters (not in the default set);  list: rocprofv3 -L
#   ba_local ba_global ba_batch            kernel stats of the local, the global and the batched local BA
#   ba_mfma_{local,global,batch}           SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES GRBM-free
#   ba_fetch_{local,global,batch} ba_write_{...}   FETCH_SIZE / WRITE_SIZE (separate passes; x2 rule for wide loads in tools/ba_pmc.py)
#   bow bow_fetch bow_write    kernel stats and FETCH_SIZE / WRITE_SIZE of the configs[3] BoW query
set -e
KEEP=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
OUT=/tmp/prof_r05          # raw rocprofv3 output stays on the box (hundreds of MB); only the summaries are copied to gpurun_out
mkdir -p $OUT $KEEP
cd /tmp && export TMPDIR=/tmp
PASSES="$@"
[ -z "$PASSES" ] && PASSES="stats sq1 sq2 fetch write grbm bow ba_local ba_global ba_batch ba_mfma_local ba_mfma_global ba_mfma_batch ba_fetch_local ba_fetch_global ba_fetch_batch ba_write_local ba_write_global ba_write_batch bow_fetch bow_write"
B="python3 $GRAFT_REPO_ROOT/bench.py"
FE="--frames 256 --steps 2 --warmup 1 --no-ba --no-global-ba --no-bow --no-inputs --cpu-sample 0 --pipelines 1"
T=$GRAFT_REPO_ROOT/tools
pmc() { local name=$1; shift; local ctrs=$1; shift; rm -rf $OUT/$name; timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1 || echo "pass $name failed"; echo $name-done; }
kst() { local name=$1; shift; rm -rf $OUT/$name; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1 || echo "pass $name failed"; echo $name-done; }
for p in $PASSES; do
  case $p in
    stats) kst stats $B --steps 5 --warmup 2 --cpu-sample 0 --pipelines 1 --no-inputs --no-ba --no-global-ba --no-bow ;;
    sq1)   pmc sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" $B $FE ;;
    sq2)   pmc sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" $B $FE ;;
    fetch) pmc fetch FETCH_SIZE $B $FE ;;
    write) pmc write WRITE_SIZE $B $FE ;;
    grbm)  pmc grbm GRBM_GUI_ACTIVE $B $FE ;;
    ta1)   pmc ta1 "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" $B $FE ;;
    tcp1)  pmc tcp1 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" $B $FE ;;
    tcc1)  pmc tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" $B $FE ;;
    list)  rocprofv3 -L > $KEEP/counters.txt 2>&1 || true ;;
    ba_local)  kst ba_local python3 $T/ba_only.py ;;
    ba_global) kst ba_global python3 $T/ba_global.py ;;
    ba_batch)  kst ba_batch python3 $T/ba_batch.py 64 ;;
    ba_mfma_local)  pmc ba_mfma_local  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES" python3 $T/ba_only.py ;;
    ba_mfma_global) pmc ba_mfma_global "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES" python3 $T/ba_global.py ;;
    ba_mfma_batch)  pmc ba_mfma_batch  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES" python3 $T/ba_batch.py 64 ;;
    ba_fetch_local)  pmc ba_fetch_local  FETCH_SIZE python3 $T/ba_only.py ;;
    ba_fetch_global) pmc ba_fetch_global FETCH_SIZE python3 $T/ba_global.py ;;
    ba_fetch_batch)  pmc ba_fetch_batch  FETCH_SIZE python3 $T/ba_batch.py 64 ;;
    ba_write_local)  pmc ba_write_local  WRITE_SIZE python3 $T/ba_only.py ;;
    ba_write_global) pmc ba_write_global WRITE_SIZE python3 $T/ba_global.py ;;
    ba_write_batch)  pmc ba_write_batch  WRITE_SIZE python3 $T/ba_batch.py 64 ;;
    bow)       kst bow python3 $T/bow_only.py ;;
    bow_fetch) pmc bow_fetch FETCH_SIZE python3 $T/bow_only.py ;;
    bow_write) pmc bow_write WRITE_SIZE python3 $T/bow_only.py ;;
    *) echo "unknown pass $p" ;;
  esac
done
# summaries -> gpurun_out
cd $GRAFT_REPO_ROOT
for d in stats bow ba_local ba_global ba_batch; do f=$(find $OUT/$d -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp $f $KEEP/${d}_kernel_stats.csv; done
python3 tools/pmc_summary.py $OUT > $KEEP/pmc_summary.json 2> $KEEP/pmc_summary.err || echo "pmc_summary failed"
python3 tools/pmc_traffic.py $OUT 128 > $KEEP/pmc_traffic.json 2> $KEEP/pmc_traffic.err || echo "pmc_traffic failed"
python3 tools/ba_pmc.py $OUT > $KEEP/ba_pmc.json 2> $KEEP/ba_pmc.err || echo "ba_pmc failed"
for l in $PASSES; do tail -c 2000 $OUT/$l.log > $KEEP/$l.log.tail 2>/dev/null || true; done
du -sh $OUT | tail -1
echo prof-done
