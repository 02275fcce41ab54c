"""Summarise rocprofv3 --pmc CSVs written by tools/prof_pmc.sh: per kernel, per counter: mean over dispatches."""
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
print(json.dumps(out, indent=1))
