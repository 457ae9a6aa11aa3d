#!/bin/bash
# The corner sweeps (value-range extremes, wide pictures) on the GPU box: bash tools/r04_sweeps2.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
tag=$1; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; timeout -k 10 900 "$@" > $out/$name.log 2>&1; echo "$name rc=$? $(tail -1 $out/$name.log | cut -c1-160)"; }
run extreme python tools/param_sweep.py 500 --gpu --extreme --seed 51
run extreme_split python tools/param_sweep.py 300 --gpu --extreme --split --seed 52
run big python tools/param_sweep.py 200 --gpu --big --seed 53
run fields_extreme python tools/param_sweep.py 300 --gpu --fields --extreme --seed 54
run plain python tools/param_sweep.py 1000 --gpu --seed 55
