R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r05v
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $out/pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-extra --no-cpu-baseline --no-parity --distinct 32 > /dev/null 2> $out/pmc.err
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$out/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
dur = collections.defaultdict(list)
for f in glob.glob("$out/pmc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
for k in dur:
    v = sorted(dur[k]); print("duration ms", k, "n", len(v), "median %.4f mean %.4f" % (v[len(v) // 2], sum(v) / len(v)))
for k in acc:
    print(k, {c: round(v / max(1, n[(k, c)]) / 1e6, 2) for c, v in acc[k].items()}, "M per launch", n[(k, "SQ_WAVES")])
PY
rm -rf $out/pmc
