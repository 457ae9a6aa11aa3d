"""Print a coarse timeline from a rocprofv3 kernel trace: every k_entropy dispatch and, per pass, the
span of the reconstruction kernels (first start .. last end), in ms relative to the first dispatch."""
import csv
import glob
import sys

paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    rows += list(csv.DictReader(open(p)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
t0 = ev[0][0]
cur = None
for s, e, k, q in ev:
    if k.startswith("k_entropy"):
        print("%9.2f .. %9.2f  %-10s q=%s  (%.1f ms)" % ((s - t0) / 1e6, (e - t0) / 1e6, k, q, (e - s) / 1e6))
rec = [(s, e, k, q) for s, e, k, q in ev if k in ("k_inter", "k_intra", "k_deblock")]
# group recon kernels into passes by gaps in k_intra count (30 per pass)
n = 0
start = None
for s, e, k, q in rec:
    if start is None:
        start = s
    if k == "k_deblock":
        n += 1
        if n % 30 == 0:
            print("recon pass %d: %9.2f .. %9.2f (%.1f ms) q=%s" % (n // 30, (start - t0) / 1e6, (e - t0) / 1e6, (e - start) / 1e6, q))
            start = None
