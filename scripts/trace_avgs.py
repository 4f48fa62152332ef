"""Average duration per (kernel, grid) of a rocprofv3 --kernel-trace CSV, in launch order of first appearance.
Usage: python scripts/trace_avgs.py <dir or kernel_trace.csv> [substring ...]   (substrings filter kernel names)"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
pats = sys.argv[2:]
f = src if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if pats and not any(p in n for p in pats):
        continue
    k = (n[:70], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
    agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, gx, gy, wg), v in agg.items():
    v2 = sorted(v)[: max(1, len(v) * 3 // 4)]                     # drop the slowest quarter (warm-up launches)
    print(f"{n:70s} grid {gx:>8s}x{gy:<5s} wg {wg:>4s}  n={len(v):4d}  avg {sum(v2)/len(v2):8.1f} us")
