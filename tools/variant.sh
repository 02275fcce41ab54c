#!/bin/bash
# A/B builds of the front end: tools/variant.sh build <name> <extra hipcc flags...> compiles orb.hip with the flags into build/var/<name>/lib.so
# (HERE: hipcc cross-compiles); tools/variant.sh run <name> [env ...] times it on the GPU box (front-end bench line, BH_LIB_PATH).
set -e
cd "$(dirname "$0")/.."
mode=$1; name=$2; shift 2
D=build/var/$name
if [ "$mode" = build ]; then
  mkdir -p $D
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function "$@" -Iinclude -c boslam_amd/csrc/orb.hip -o $D/orb.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $D/lib.so $D/orb.o $(ls boslam_amd/csrc/*.o | grep -v /orb.o)
else
  L=$PWD/$D/lib.so; [ "$name" = base ] && L=$PWD/boslam_amd/csrc/libboslam_hip.so
  env "$@" BH_LIB_PATH=$L python bench.py --steps 10 --warmup 3 --no-ba --no-global-ba --no-bow --no-inputs --cpu-sample 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name $*', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
fi
