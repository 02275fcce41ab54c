#!/bin/bash
# Dense-Cholesky iteration loop on the GPU box: every test that solves a reduced camera system, then the local (configs[2]) and global (configs[4]) lines.
set -o pipefail
T=$1
timeout -k 10 900 python -m pytest tests/test_ba_gpu.py tests/test_band_solve_gpu.py tests/test_global_ba_gpu.py tests/test_threads_gpu.py -x -q -m gpu > gpurun_out/chol_$T.test.log 2>&1 || { tail -40 gpurun_out/chol_$T.test.log; exit 1; }
tail -2 gpurun_out/chol_$T.test.log
timeout -k 10 300 python tools/ba_only.py > gpurun_out/chol_$T.local.json 2> gpurun_out/chol_$T.err || { tail -20 gpurun_out/chol_$T.err; exit 1; }
timeout -k 10 300 python tools/ba_global.py > gpurun_out/chol_$T.global.json 2>> gpurun_out/chol_$T.err || { tail -20 gpurun_out/chol_$T.err; exit 1; }
python - <<PY
import json
for w in ("local", "global"):
    t=open("gpurun_out/chol_$T.%s.json" % w).read()
    d=json.loads(t[t.index("{"):])
    print(w, d["value"], d.get("ms_per_iter"), d.get("kernel_ms_per_iter"))
PY
