"""Coarse timeline from a rocprofv3 kernel trace (any batch shape): every k_entropy* / k_dbprep dispatch, and the busy spans
of the reconstruction kernels (dispatches less than 0.5 ms apart merged), in ms relative to the first dispatch.
Usage: python tools/prof_timeline2.py <trace dir> [from_ms]"""
import csv
import glob
import sys

rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
t0 = ev[0][0]
out = []
span = None
for s, e, k, q in ev:
    if k.startswith("k_entropy") or k.startswith("k_dbprep"):
        out.append((s, "%9.2f .. %9.2f  %-12s q=%s  (%.1f ms)" % ((s - t0) / 1e6, (e - t0) / 1e6, k, q, (e - s) / 1e6)))
    elif k.startswith("k_"):
        if span and s - span[1] < 500000 and q == span[3]:
            span[1] = max(span[1], e)
            span[2] += 1
        else:
            if span:
                out.append((span[0], "%9.2f .. %9.2f  recon x%-5d q=%s  (%.1f ms)" % ((span[0] - t0) / 1e6, (span[1] - t0) / 1e6, span[2], span[3], (span[1] - span[0]) / 1e6)))
            span = [s, e, 1, q]
if span:
    out.append((span[0], "%9.2f .. %9.2f  recon x%-5d q=%s  (%.1f ms)" % ((span[0] - t0) / 1e6, (span[1] - t0) / 1e6, span[2], span[3], (span[1] - span[0]) / 1e6)))
for s, line in sorted(out):
    if (s - t0) / 1e6 >= lo:
        print(line)
