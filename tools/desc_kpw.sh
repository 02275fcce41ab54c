#!/bin/bash
# Diagnostic: k_describe with 4 / 8 / 16 frames per wave (DS_KPW) - built HERE, timed on the GPU box with BH_LIB_PATH.
# build: bash tools/desc_kpw.sh build   run (gpurun): bash tools/desc_kpw.sh run
set -e
cd "$(dirname "$0")/.."
D=build/desc_kpw
if [ "$1" = build ]; then
  mkdir -p $D
  for n in ${KPW:-4 8}; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DDS_KPW=$n -Iinclude -c boslam_amd/csrc/orb.hip -o $D/orb_$n.o 2>/dev/null
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $D/lib_$n.so $D/orb_$n.o $(ls boslam_amd/csrc/*.o | grep -v /orb.o)
  done
else
  for n in ${KPW:-4 8}; do
    BH_LIB_PATH=$PWD/$D/lib_$n.so timeout -k 10 200 python -m pytest tests/test_orb_gpu.py -x -q -m gpu 2>&1 | tail -1
    BH_LIB_PATH=$PWD/$D/lib_$n.so python bench.py --steps 6 --warmup 2 --no-ba --no-global-ba --no-bow --no-inputs --cpu-sample 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DS_KPW $n: k_describe', d['roofline']['kernel_ms_per_step'].get('k_describe'), 'fps', d['value'])"
  done
fi
