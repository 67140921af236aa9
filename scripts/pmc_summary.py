"""Condense rocprofv3 --pmc counter_collection CSVs (one directory per pass) into
profiles/<name>_pmc_summary.csv: per kernel and counter the launches, mean, median, max.
usage: python scripts/pmc_summary.py out.csv pass_dir [pass_dir ...]"""
import csv
import glob
import statistics
import sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
rows = []
for d in dirs:
    acc = defaultdict(list)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        rows.append((k, c, len(v), sum(v) / len(v), statistics.median(v), max(v)))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "mean", "median", "max"])
    w.writerows(rows)
print(f"{len(rows)} rows -> {out}")
