#!/bin/bash
# throughput of the front-end for a few chunk sizes / pipeline counts (tuning aid)
for p in ${PIPES:-2 3 4}; do for c in ${CHUNKS:-64 96 128 192 256}; do
  v=$(BH_ORB_CHUNK=$c python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-ba --no-global-ba --no-bow --pipelines $p 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
  echo "pipelines=$p chunk=$c fps=$v"
done; done
