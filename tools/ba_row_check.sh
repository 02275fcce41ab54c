#!/bin/bash
# Global-BA iteration loop on the GPU box: the sharded / row-form tests, then the configs[4] line.   usage: bash tools/ba_row_check.sh <tag>
set -o pipefail
T=$1
timeout -k 10 700 python -m pytest tests/test_global_ba_gpu.py -x -q -m gpu > gpurun_out/bag_$T.test.log 2>&1 || { tail -30 gpurun_out/bag_$T.test.log; exit 1; }
tail -2 gpurun_out/bag_$T.test.log
timeout -k 10 300 python tools/ba_global.py > gpurun_out/bag_$T.json 2> gpurun_out/bag_$T.err || { tail -20 gpurun_out/bag_$T.err; exit 1; }
python - <<PY
import json
t=open("gpurun_out/bag_$T.json").read()
d=json.loads(t[t.index("{"):])
print(d["value"], d["ms_per_iter"], d["kernel_ms_per_iter"], d["robust_chi2"][-1])
PY
