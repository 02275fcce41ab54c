"""Summarise the bundle-adjustment PMC passes of tools/prof_r04.sh -> profiles/r04_ba_pmc.json.

Per workload (local = configs[2] one window, global = configs[4], batch = 64 windows) and kernel: dispatches, mean duration from
the pass's own kernel trace, mean counter values per dispatch, and
  mfma_f64_flops     = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512        (the counter's unit is 512 operations; checked against the
                       analytic count of k_ba_chol_dense: one v_mfma_f64_16x16x4 = 2 048 flop = 4 counts)
  mfma_tflops        = mfma_f64_flops / duration;  mfma_frac_of_peak = that / 78.6 TF (f64 matrix peak)
  mfma_busy_simds    = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz): how many SIMDs' matrix pipes were busy on average (the counter
                       is in cycles, summed over the chip: 64 per v_mfma_f64_16x16x4); a one-workgroup kernel can reach 4 (its CU)
  mfma_busy_frac_chip = that / 1024 SIMDs
  hbm_read_bytes     = FETCH_SIZE KiB x 1024 x 2   (gfx950: FETCH_SIZE reports half the bytes of >= 8 B/lane coalesced reads, MI355X
                       guide, HBM / rocprofv3 section; every BA kernel reads f64 or wider)
  hbm_write_bytes    = WRITE_SIZE KiB x 1024
  hbm_gbps           = (read + write) / duration
Usage: python tools/ba_pmc.py <raw rocprofv3 root>"""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
PEAK_F64_MFMA = 78.6e12


def load(passdir):
    ctr = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
    for path in glob.glob(os.path.join(root, passdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            ctr[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for path in glob.glob(os.path.join(root, passdir, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            dur[k].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    return ctr, dur


out = {"_note": __doc__.split("Usage")[0].strip()}
prev = json.load(open(root)) if os.path.isfile(root) else None        # a summary written earlier: derive again from its counters
for wl in ("local", "global", "batch"):
    res = {}
    per = {}
    if prev is not None:
        for k, v in prev.get(wl, {}).items():
            per[k] = {"counters": dict(v["counters"]), "ns": v["mean_us"] * 1e3, "dispatches": v["dispatches"]}
    for kind in (() if prev is not None else ("mfma", "fetch", "write")):
        ctr, dur = load(f"ba_{kind}_{wl}")
        for k, cs in ctr.items():
            if not (k.startswith("k_ba") or k.startswith("k_bcr") or k.startswith("k_bb") or k.startswith("k_b2") or k.startswith("k_pose")):
                continue
            e = per.setdefault(k, {"counters": {}, "ns": None, "dispatches": 0})
            for c, v in cs.items():
                e["counters"][c] = sum(v) / len(v)
            e["dispatches"] = max(e["dispatches"], max(len(v) for v in cs.values()))
            if kind == "mfma" and dur.get(k):
                e["ns"] = sum(dur[k]) / len(dur[k])
            elif e["ns"] is None and dur.get(k):
                e["ns"] = sum(dur[k]) / len(dur[k])
    for k, e in sorted(per.items()):
        c = e["counters"]; ns = e["ns"] or 0.0
        d = {}
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c:
            fl = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
            d["mfma_f64_flops"] = fl
            if ns > 0:
                d["mfma_tflops"] = round(fl / ns * 1e-3, 5); d["mfma_frac_of_peak"] = round(fl / (ns * 1e-9) / PEAK_F64_MFMA, 6)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and ns > 0:
            d["mfma_busy_simds"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (ns * 2.4), 4)
            d["mfma_busy_frac_chip"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (ns * 2.4) / 1024.0, 6)
        if "SQ_INSTS_VALU" in c and c.get("SQ_WAVES"):
            d["valu_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        rb = c.get("FETCH_SIZE"); wb = c.get("WRITE_SIZE")
        if rb is not None: d["hbm_read_bytes"] = rb * 1024.0 * 2.0; d["hbm_read_bytes_uncorrected"] = rb * 1024.0
        if wb is not None: d["hbm_write_bytes"] = wb * 1024.0
        if ns > 0 and rb is not None and wb is not None:
            d["hbm_gbps"] = round((rb * 2048.0 + wb * 1024.0) / ns, 3)
            d["hbm_gbps_uncorrected"] = round((rb * 1024.0 + wb * 1024.0) / ns, 3)      # above 8 000 corrected: the x2 rule does not hold for that kernel
        res[k] = {"dispatches": e["dispatches"], "mean_us": round(ns * 1e-3, 3), "counters": {a: round(b, 1) for a, b in sorted(c.items())}, "derived": d}
    out[wl] = res
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from boslam_amd import _profiles
out["_meta"] = _profiles.meta(["ba.hip"])
print(json.dumps(out, indent=1))
