#!/bin/bash
# Usage: bash tools/b_timeline_run.sh <tag> [streams] [frames]  -- kernel-trace timeline of pipelined passes over an I B B P batch
tag=$1; S=${2:-32}; F=${3:-30}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
H264MI_SLICE_STATS=0 timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 $R/tools/b_profile.py $S $F > $out/out.log 2> $out/err.log || tail -5 $out/err.log
tail -3 $out/err.log
python3 $R/tools/prof_timeline2.py $out/tr > $out/timeline.txt
tail -60 $out/timeline.txt
