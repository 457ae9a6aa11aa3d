"""Summarise a rocprofv3 --kernel-trace CSV: per kernel count / total / avg / min / max (us) and
the first dispatches of each kernel (to separate I-picture launches from P-picture launches)."""
import csv
import glob
import sys
from collections import defaultdict

paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not paths:
    sys.exit("no kernel_trace.csv under " + sys.argv[1])
rows = []
for p in paths:
    with open(p) as f:
        rows += list(csv.DictReader(f))
by = defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print("%-40s %8s %12s %10s %10s %10s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us"))
for k, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    d = [(e - s) / 1e3 for s, e in v]
    print("%-40s %8d %12.3f %10.1f %10.1f %10.1f" % (k[:40], len(d), sum(d) / 1e3, sum(d) / len(d), min(d), max(d)))
for k, v in by.items():
    v.sort()
    print(k[:40], "first dispatches (us):", [round((e - s) / 1e3, 1) for s, e in v[:6]])
