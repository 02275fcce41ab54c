#!/bin/bash
# Front-end iteration loop on the GPU box: the bit-exactness tests of the front end, then a front-end-only bench line.
# usage (from the repo root, inside gpurun): bash tools/fe_iter.sh <tag>
set -o pipefail
T=$1
timeout -k 10 500 python -m pytest tests/test_orb_gpu.py tests/test_edge_gpu.py -x -q -m gpu > gpurun_out/fe_$T.test.log 2>&1 || { tail -30 gpurun_out/fe_$T.test.log; exit 1; }
tail -2 gpurun_out/fe_$T.test.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-ba --no-global-ba --no-bow --cpu-sample 0 > gpurun_out/fe_$T.json 2> gpurun_out/fe_$T.err || { tail -20 gpurun_out/fe_$T.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/fe_$T.json").read().strip().splitlines()[-1])
print("fps", d["value"], "ms/step", d["ms_per_step"], "parity", d.get("parity_sample_ok"))
print(d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"])
print({k: v["frames_per_s"] for k, v in d.get("inputs", {}).items()})
PY
