#!/bin/bash
# Builder iteration loop on the GPU box: every test that goes through bh_ba_set_problem / the BoW set-up, then the configs[4] line (set_problem_s).
set -o pipefail
T=$1
timeout -k 10 900 python -m pytest tests/test_ba_gpu.py tests/test_global_ba_gpu.py tests/test_bow_gpu.py tests/test_window_gpu.py tests/test_band_solve_gpu.py -x -q -m gpu > gpurun_out/bld_$T.test.log 2>&1 || { tail -40 gpurun_out/bld_$T.test.log; exit 1; }
tail -2 gpurun_out/bld_$T.test.log
timeout -k 10 300 python tools/ba_global.py > gpurun_out/bld_$T.json 2> gpurun_out/bld_$T.err || { tail -20 gpurun_out/bld_$T.err; exit 1; }
python - <<PY
import json
t=open("gpurun_out/bld_$T.json").read()
d=json.loads(t[t.index("{"):])
print(d["value"], d["ms_per_iter"], "set_problem_s", d["set_problem_s"], d["robust_chi2"][-1])
PY
