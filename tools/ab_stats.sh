#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of prebuilt variant libraries: bash tools/ab_stats.sh <tag> "" <lib> ...   ("" = the default build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename "${lib:-default}" .so)
  out=$R/gpurun_out/$tag/$name
  mkdir -p $out
  H264MI_LIB=${lib:+$R/$lib} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --distinct 32 --no-extra --no-cpu-baseline --no-parity --steps 3 > $out.json 2> $out.err || tail -3 $out.err
  echo "== $name"
  python3 - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("k_"):
            print("  %-14s calls %5s avg %10.1f us  total %9.2f ms" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
