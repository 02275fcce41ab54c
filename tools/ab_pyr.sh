#!/bin/bash
# A/B of the two pyramid kernels on the front-end bench line (same box, alternating): BH_ORB_PYR_VALU=1 selects k_pyr_down_bits
for i in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export BH_ORB_PYR_VALU=1; else unset BH_ORB_PYR_VALU; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-ba --no-global-ba --no-bow --no-inputs --cpu-sample 0 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
    python - $v <<PY
import json,sys
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("valu_kernel" if sys.argv[1]=="1" else "mfma_kernel", "fps", d["value"], d["roofline"]["kernel_ms_per_step"])
PY
  done
done
