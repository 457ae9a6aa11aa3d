#!/bin/bash
# Usage: bash tools/timeline_run.sh <tag> [bench args]  -- kernel-trace timeline of the pipelined passes
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 $R/bench.py --no-cpu-baseline --no-parity "$@" > $out/bench.json 2> $out/err.log || tail -5 $out/err.log
python3 -c "import json; d=json.load(open('$out/bench.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
python3 $R/tools/prof_timeline.py $out/tr
