"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (steady-state steps only): total idle time per
step and the (previous kernel -> next kernel) pairs that accumulate most of it.
Usage: python scripts/trace_gaps.py <dir or kernel_trace.csv> [min_gap_us]"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
f = src if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "nll_loss_forward" in r["Kernel_Name"] or "log_softmax_forward" in r["Kernel_Name"]]
marks = [m for j, m in enumerate(marks) if j == 0 or m - marks[j - 1] > 50]
if len(marks) < 3:
    sys.exit("not enough steps in the trace")
lo, hi = marks[-3], marks[-1]
steps = 2
seg = rows[lo:hi]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
pairs = collections.defaultdict(lambda: [0, 0.0])
for a, b in zip(seg[:-1], seg[1:]):
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if gap >= min_gap:
        k = (a["Kernel_Name"][:48], b["Kernel_Name"][:48])
        pairs[k][0] += 1
        pairs[k][1] += gap
print(f"{len(seg) / steps:.0f} kernels/step, busy {busy / steps / 1e3:.2f} ms/step, span {span / steps / 1e3:.2f} ms/step, "
      f"idle {(span - busy) / steps / 1e3:.2f} ms/step; gaps >= {min_gap} us: {sum(v[1] for v in pairs.values()) / steps / 1e3:.2f} ms/step")
for (a, b), (n, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t / steps:8.1f} us/step  x{n / steps:5.1f}  {a:48s} -> {b}")
