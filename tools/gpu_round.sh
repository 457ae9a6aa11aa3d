#!/bin/bash
# One GPU-box call: parity tests, smoke, default bench line, rocprofv3 kernel stats of the same command.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag>
set -o pipefail
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/${tag}_pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/${tag}_smoke.log
timeout -k 10 500 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"; tail -1 $out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o ${tag} -- python3 $R/bench.py --steps 3 --no-cpu-baseline --no-extra > $out/${tag}_prof_bench.json 2> $out/${tag}_prof.err; echo "rocprof rc=$?"
ls $out/${tag}_prof | head
