"""Summarise rocprofv3 --pmc CSVs written by tools/prof_pmc.sh: per kernel, per counter: mean over dispatches.

Derived figures (key "derived"), with the formula each one is:
  valu_issue_util   = SQ_ACTIVE_INST_VALU / (32 x GRBM_GUI_ACTIVE).  SQ_ACTIVE_INST_* count quad-cycles summed over every SIMD of
                      the chip; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the launch lasted GRBM / 8 cycles and the chip's
                      256 CUs x 4 SIMDs offered 1024 x GRBM / 8 cycles = 32 x GRBM quad-cycles of vector issue.
  salu_per_valu     = SQ_INSTS_SALU / SQ_INSTS_VALU (wave-instructions).
  valu_per_wave, salu_per_wave, lds_per_wave = SQ_INSTS_* / SQ_WAVES.
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS (extra LDS-array cycles over LDS-instruction-active cycles;
                      can exceed 1 when most accesses are many-way conflicts).
  wait_frac         = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of wave lifetime parked at s_waitcnt / s_barrier)."""
import csv, glob, os, sys, json
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0]
        k = k.replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    if not k.startswith("k_"):
        continue
    out[k] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
    c = out[k]; dv = {}
    def ratio(a, b, scale=1.0):
        return round(c[a] / (scale * c[b]), 4) if a in c and b in c and c[b] else None
    dv["valu_issue_util"] = ratio("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", 32.0)
    dv["salu_per_valu"] = ratio("SQ_INSTS_SALU", "SQ_INSTS_VALU")
    dv["valu_per_wave"] = ratio("SQ_INSTS_VALU", "SQ_WAVES"); dv["salu_per_wave"] = ratio("SQ_INSTS_SALU", "SQ_WAVES")
    dv["lds_per_wave"] = ratio("SQ_INSTS_LDS", "SQ_WAVES")
    dv["lds_conflict_frac"] = ratio("SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS")
    dv["wait_frac"] = ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
    out[k]["derived"] = {a: b for a, b in dv.items() if b is not None}
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from boslam_amd import _profiles
out["_meta"] = _profiles.meta(["orb.hip", "match.hip"])
print(json.dumps(out, indent=1))
