#!/bin/bash
# Usage: bash tools/variant_run.sh "<EXTRA flags>" [bench args] -- rebuild the library with EXTRA and run slice stats + bench
extra=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rm -f h264decode_amd/csrc/_build/*.o
make -s -j16 -C h264decode_amd/csrc EXTRA="$extra" 2>&1 | grep -E "error" 
echo "== $extra"
timeout -k 10 200 python tools/slice_stats.py 1 4 2>&1 | grep "^slice [01] "
timeout -k 10 400 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
