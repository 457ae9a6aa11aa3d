#!/bin/bash
# HBM traffic per kernel launch of the bench workload: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
# do not fit one pass), corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950 for wide
# coalesced reads; both counters are in KiB).  Writes profiles/pmc_traffic.json + the raw per-kernel table.
# Usage (GPU box, repo root): bash tools/pmc_traffic.sh <tag> [bench args, e.g. --streams 128]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity "$@" > $out/$c.json 2> $out/$c.err || tail -3 $out/$c.err
done
python3 $R/tools/pmc_traffic_summary.py $out "$@"
