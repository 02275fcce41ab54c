"""profiles/pmc_traffic.json from a tools/prof_pmc.sh run: HBM bytes PER FRAME for every front-end kernel.

FETCH_SIZE / WRITE_SIZE are in KiB (MI355X guide, HBM/rocprofv3 section).  gfx950 correction: FETCH_SIZE under-reports by 2x
for kernels whose global loads are >= 8 B per lane (calibrated on k_gray_mask, whose input bytes are known exactly); kernels
with dword-or-narrower loads are taken as reported.  Usage: python tools/pmc_traffic.py gpurun_out/<pmc dir> <frames per launch>"""
import json, subprocess, sys, os

root, fpl = sys.argv[1], int(sys.argv[2])
here = os.path.dirname(os.path.abspath(__file__))
d = json.loads(subprocess.run([sys.executable, os.path.join(here, "pmc_summary.py"), root], capture_output=True, text=True, check=True).stdout)
WIDE = {"k_gray_mask": 2, "k_gray_bits": 2, "k_gray_l1": 2, "k_pyr_down": 2, "k_pyr_down_bits": 2, "k_select": 2, "k_bow_scan": 2}      # >= 8 B per lane loads (12 B + 8 B, 12 B, 8 B records, 16 B)
PER_LAUNCH = {"k_bow_scan", "k_bow_topk", "k_bow_topk_slices", "k_bow_pack", "k_inv_gather", "k_inv_finish", "k_inv_count", "k_inv_fill", "k_inv_widen"}                # reported per launch (= per query), not per frame
out = {"_note": "HBM bytes per FRAME = (FETCH_SIZE x factor + WRITE_SIZE) KiB per launch / frames per launch; rocprofv3 --pmc, FETCH_SIZE and "
                "WRITE_SIZE in separate passes, bench.py --frames 256 --pipelines 1 (launches of %d frames); k_pyr_down and k_blur are "
                "summed over their launches per frame batch; k_bow_* entries are bytes per launch (one query of configs[3] on one GPU).  factor 2 = kernels with >= 8 B/lane loads (gfx950 FETCH_SIZE correction, "
                "calibrated on k_gray_mask), 1 otherwise." % fpl, "_factors": WIDE}
agg = {}
out["_meta"] = d.get("_meta")
for k, v in d.items():
    if k.startswith("_"):
        continue
    base = k.split("<")[0]
    if base.startswith("k_ba") or "FETCH_SIZE" not in v:
        continue
    per_launch = (v["FETCH_SIZE"] * WIDE.get(base, 1) + v.get("WRITE_SIZE", 0.0)) * 1024.0
    # the pyramid runs one launch per level transition (7 of them; 6 when k_gray_l1 makes level 1): counted from the dispatches
    launches_per_batch = 1
    if base in ("k_pyr_down", "k_pyr_down_bits"):
        nf = d.get("k_fast", {}).get("dispatches", 0)
        launches_per_batch = round(v.get("dispatches", 0) / nf) if nf else 7
    agg[base] = agg.get(base, 0.0) + per_launch * launches_per_batch / (1 if base in PER_LAUNCH else fpl)
for k, v in sorted(agg.items()):
    out[k] = int(round(v))
print(json.dumps(out, indent=1))
