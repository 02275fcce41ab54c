#!/bin/bash
# Diagnostic: k_fast cut after phase N (N = 1..4: tile load, pretest, ring test, blur) - built HERE (hipcc cross-compiles), timed on the GPU box
# with BH_LIB_PATH.  Results are wrong by construction; only kernel_ms_per_step.k_fast is read.
# build:  bash tools/fast_phases.sh build      run (gpurun):  bash tools/fast_phases.sh run
set -e
cd "$(dirname "$0")/.."
D=build/fast_phases
if [ "$1" = build ]; then
  mkdir -p $D
  for n in ${PHASES:-1 2 3 4}; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DFAST_STOP=$n -Iinclude -c boslam_amd/csrc/orb.hip -o $D/orb_$n.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $D/lib_$n.so $D/orb_$n.o $(ls boslam_amd/csrc/*.o | grep -v /orb.o)
  done
else
  for n in ${PHASES:-1 2 3 4}; do
    BH_LIB_PATH=$PWD/$D/lib_$n.so python bench.py --steps 6 --warmup 2 --no-ba --no-global-ba --no-bow --no-inputs --cpu-sample 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stop after phase $n: k_fast', d['roofline']['kernel_ms_per_step'].get('k_fast'))"
  done
fi
