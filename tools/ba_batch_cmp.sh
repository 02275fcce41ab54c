#!/bin/bash
# batched / single local BA with and without the wave-per-pair Schur rule (GPU box)
for v in 512 0; do
  echo "BH_BA_SCHUR_WAVE_AVG=$v"
  BH_BA_SCHUR_WAVE_AVG=$v python tools/ba_batch.py 64 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print({k: d[k] for k in ('value', 'ms_per_round', 'ms_per_iter', 'speedup_vs_one_at_a_time', 'single_window_iters_per_s', 'value_incl_set_problem', 'set_problem_ms', 'speedup_incl_set_problem') if k in d}, d.get('workload', '')[:40])
"
done
