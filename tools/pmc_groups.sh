#!/bin/bash
# Counter passes over the bench workload, one rocprofv3 run per group (a group that names a counter this ROCm does not know fails alone):
#   bash tools/pmc_groups.sh <tag> "<bench args>" "<group 1>" "<group 2>" ...     (GPU box, repo root; --kernel-trace + --pmc only)
tag=$1; bargs=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for g in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $out/g$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-extra $bargs > $out/g$i.json 2> $out/g$i.err || { echo "group $i ($g) failed:"; tail -2 $out/g$i.err; }
done
python3 - $out <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(acc):
    if k.startswith("k_"):
        print(k, " ".join("%s=%.4g/launch(%d)" % (c, acc[k][c] / max(1, len(n[k][c])), len(n[k][c])) for c in sorted(acc[k])))
PY
