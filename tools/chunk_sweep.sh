#!/bin/bash
# Front-end step time against the chunk size (frames per launch) and the number of chunk pipelines.   usage (gpurun): bash tools/chunk_sweep.sh "chunks" "pipes" [repeats]
for r in $(seq 1 ${3:-1}); do
for c in ${1:-64 96 128 160 200 256}; do
  for p in ${2:-2 4}; do
    BH_ORB_CHUNK=$c BH_ORB_PIPELINES=$p python bench.py --steps 20 --warmup 3 --no-ba --no-global-ba --no-bow --no-inputs --cpu-sample 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $c pipes $p: fps', d['value'], 'ms', d['ms_per_step'])"
  done
done
done
