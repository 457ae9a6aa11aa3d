#!/bin/bash
# K4's instruction budget by part (GPU box): the -DK4_EXP variants leave a part out (wrong pictures: parity is off) -- kernel time from the bench line,
# vector instructions per launch from one --pmc pass each.   bash tools/k4_budget.sh <tag>
tag=${1:-k4b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in "" k4e1 k4e2 k4e4 k4e7; do
  lib=$R/h264decode_amd/libh264mi${v:+_$v}.so
  H264MI_LIB=$lib timeout -k 10 300 python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-parity --distinct 32 > $out/bench_$v.json 2> $out/bench_$v.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_$v.json").read().strip().splitlines()[-1])
print("variant '$v': k_inter ms", d["roofline"]["per_launch"]["k_inter"]["ms"], "inter per step", d["roofline"]["all_kernels_ms_per_step"]["inter"])
PY
  H264MI_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $out/pmc_$v -- python3 $R/bench.py --steps 1 --warmup 0 --no-extra --no-cpu-baseline --no-parity --distinct 32 > /dev/null 2> $out/pmc_$v.err
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$out/pmc_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_inter"):
            acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(r["Kernel_Name"], r["Counter_Name"])] += 1
for k in acc:
    print("   ", k, {c: round(v / max(1, n[(k, c)]) / 1e6, 1) for c, v in acc[k].items()}, "M per launch")
PY
  rm -rf $out/pmc_$v
done
