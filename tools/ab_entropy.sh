#!/bin/bash
# A/B two versions of k_entropy.hip in one GPU-box call: bash tools/ab_entropy.sh a.hip b.hip [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
a=$1; b=$2; shift 2
cd $R
for rep in 1 2; do
  for v in $a $b; do
    cp $v h264decode_amd/csrc/k_entropy.hip
    make -s -j16 -C h264decode_amd/csrc 2>&1 | grep error
    echo -n "$(basename $v) rep $rep: "
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']['entropy'])"
  done
done
