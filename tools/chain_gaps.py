"""From a rocprofv3 kernel trace: durations of the reconstruction kernels and the idle gaps between consecutive ones
(single-stream chain K4 -> K3 -> K5 per picture).  Usage: chain_gaps.py <trace dir>"""
import csv, glob, sys
import numpy as np
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
rec = [(s, e, k) for s, e, k in ev if k.split("(")[0] in ("k_inter", "k_intra", "k_deblock", "k_intra_x", "k_deblock_x", "k_inter_b")]
dur, gaps = {}, []
for i, (s, e, k) in enumerate(rec):
    dur.setdefault(k.split("(")[0], []).append((e - s) / 1e3)
    if i and s - rec[i - 1][1] < 5e6:
        gaps.append((s - rec[i - 1][1]) / 1e3)
for k, v in dur.items():
    print("%-12s n=%5d  mean %8.1f us  median %8.1f us" % (k, len(v), np.mean(v), np.median(v)))
print("gaps between consecutive reconstruction kernels: n=%d mean %.1f us median %.1f us p90 %.1f us" % (len(gaps), np.mean(gaps), np.median(gaps), np.percentile(gaps, 90)))
span = (rec[-1][1] - rec[0][0]) / 1e6
print("span %.1f ms, kernels %.1f ms, gaps %.1f ms" % (span, sum(sum(v) for v in dur.values()) / 1e3, sum(gaps) / 1e3))
