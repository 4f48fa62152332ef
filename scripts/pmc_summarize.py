"""Summarise rocprofv3 counter_collection CSVs: mean of every counter per (kernel, grid size)."""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (r["Kernel_Name"][:60], r.get("Grid_Size", ""))
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(rows):
    vals = {k: sum(v) / len(v) for k, v in rows[key].items()}
    print(key[0], "grid", key[1], "n", len(next(iter(rows[key].values()))))
    print("   " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(vals.items())))
