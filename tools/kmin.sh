#!/bin/bash
# Minimum (uncontended) duration of every kernel of the bench workload: rocprofv3 --kernel-trace of bench.py --no-extra --steps 3.  Usage: bash tools/kmin.sh <tag> [lib]
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; lib=$2
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
H264MI_LIB=${lib:+$R/$lib} timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 $R/bench.py --distinct 32 --no-extra --no-cpu-baseline --no-parity --steps 3 > $out/bench.json 2> $out/err.log || tail -3 $out/err.log
python3 - $out/tr <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print({k: (round(min(x), 3), round(sorted(x)[len(x) // 2], 3), len(x)) for k, x in sorted(d.items()) if k.startswith("k_")})
PY
