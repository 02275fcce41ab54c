"""How well do the chunk pipelines of bh_orb_detect_describe overlap?  Reads a rocprofv3 --kernel-trace CSV of the front-end bench and
prints, for the last `steps` worth of k_* dispatches: wall span, sum of kernel durations, the share of the span with 0 / 1 / 2 / 3+
kernels in flight, and per kernel the time during which it ran alone.   usage: python tools/overlap.py <kernel_trace.csv> [tail_fraction]"""
import csv, sys, json
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("k_"): continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1 - frac)):]          # steady state: the last part of the run
t0 = min(r[0] for r in rows); t1 = max(r[1] for r in rows)
ev = []
for s, e, k in rows: ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
active = defaultdict(int); n = 0; last = t0
hist = defaultdict(int); alone = defaultdict(int); pair = defaultdict(int)
for t, d, k in ev:
    dt = t - last
    if dt > 0:
        hist[min(n, 3)] += dt
        names = sorted(a for a, c in active.items() if c > 0)
        if n == 1: alone[names[0]] += dt
        elif n >= 2: pair["+".join(names)] += dt
    active[k] += d; n += d; last = t
span = t1 - t0
dur = defaultdict(int); cnt = defaultdict(int)
for s, e, k in rows: dur[k] += e - s; cnt[k] += 1
out = {"span_ms": span / 1e6, "sum_kernel_ms": sum(dur.values()) / 1e6,
       "in_flight_share": {str(k): round(v / span, 4) for k, v in sorted(hist.items())},
       "per_kernel": {k: {"calls": cnt[k], "total_ms": round(dur[k] / 1e6, 3), "avg_us": round(dur[k] / cnt[k] / 1e3, 2), "alone_ms": round(alone[k] / 1e6, 3)} for k in sorted(dur)},
       "top_overlaps_ms": {k: round(v / 1e6, 3) for k, v in sorted(pair.items(), key=lambda x: -x[1])[:12]}}
print(json.dumps(out, indent=1))
