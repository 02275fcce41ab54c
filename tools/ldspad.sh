#!/bin/bash
for n in 0 1536 4096 0 1536 4096; do
  if [ $n = 0 ]; then unset BH_LIB_PATH; else export BH_LIB_PATH=$PWD/build/ldspad/lib_$n.so; fi
  python bench.py --steps 20 --warmup 3 --no-ba --no-global-ba --no-bow --no-inputs --cpu-sample 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pad $n words: fps', d['value'], d['roofline']['kernel_ms_per_step'])"
done
