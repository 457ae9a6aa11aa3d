#!/bin/bash
# A/B of an environment switch on the GPU box: bash tools/ab_env.sh <tag> VAR val1 val2 ...   (parity matrix once per value, then two bench runs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
tag=$1; var=$2; shift 2
mkdir -p gpurun_out/$tag
for val in "$@"; do
  env $var=$val timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "matrix or 1080p or c5_share" > gpurun_out/$tag/$val.pytest.log 2>&1 || { echo "$var=$val: PARITY FAILED"; tail -5 gpurun_out/$tag/$val.pytest.log; continue; }
  for rep in 1 2; do
    echo -n "$var=$val rep $rep: "
    env $var=$val timeout -k 10 300 python bench.py --distinct 32 --no-extra --no-cpu-baseline --no-parity --steps 5 2>gpurun_out/$tag/$val.err | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
  done
done
