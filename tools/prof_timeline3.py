"""Pass-level timeline from a rocprofv3 kernel trace of the bench workload (I P P P ... batches: 89 reconstruction launches per pass): every
k_entropy / k_dbprep dispatch and, per pass, the span of its reconstruction launches and the time they were actually running.
Usage: python tools/prof_timeline3.py <trace dir> [launches per pass]"""
import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 89
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
t0 = ev[0][0]
out = []
rec = [(s, e, k) for s, e, k in ev if k in ("k_inter", "k_intra", "k_deblock", "k_inter_b")]
for i in range(0, len(rec), per):
    ch = rec[i:i + per]
    busy = sum(e - s for s, e, _ in ch)
    out.append((ch[0][0], "%9.2f .. %9.2f  recon pass %2d (%d launches): span %.1f ms, kernels busy %.1f ms" % ((ch[0][0] - t0) / 1e6, (ch[-1][1] - t0) / 1e6, i // per, len(ch), (ch[-1][1] - ch[0][0]) / 1e6, busy / 1e6)))
for s, e, k in ev:
    if k.startswith("k_entropy") or k.startswith("k_dbprep") or k.startswith("k_pack"):
        out.append((s, "%9.2f .. %9.2f  %-10s (%.1f ms)" % ((s - t0) / 1e6, (e - t0) / 1e6, k, (e - s) / 1e6)))
for _, line in sorted(out):
    print(line)
