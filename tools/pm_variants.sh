#!/bin/bash
# diagnostic: where does k_pyr_mfma's time go?  Rebuilds orb.o with parts of the kernel compiled out (wrong results) and times the front end.
set -o pipefail
cd boslam_amd/csrc
VARIANTS=("" "-DPM_NOTAPS" "-DPM_NOTAPS -DPM_NOMASK -DPM_NOFIX" "-DPM_NOMASK -DPM_NOFIX")
for v in "${VARIANTS[@]}"; do
  rm -f orb.o; make -s FLAGS_orb="$v" 2>/dev/null >/dev/null || { echo build failed; exit 1; }
  cd ../..
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-ba --no-global-ba --no-bow --no-inputs --cpu-sample 0 > gpurun_out/pmv.json 2> gpurun_out/pmv.err || { tail -5 gpurun_out/pmv.err; exit 1; }
  python - "$v" <<PY
import json,sys
d=json.loads(open("gpurun_out/pmv.json").read().strip().splitlines()[-1])
print(repr(sys.argv[1]), "fps", d["value"], d["roofline"]["kernel_ms_per_step"])
PY
  cd boslam_amd/csrc
done
