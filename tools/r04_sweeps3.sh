#!/bin/bash
# A last broad pass over every sweep mode with fresh seeds: bash tools/r04_sweeps3.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
tag=$1; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; timeout -k 10 900 "$@" > $out/$name.log 2>&1; echo "$name rc=$? $(tail -1 $out/$name.log | cut -c1-160)"; }
run plain python tools/param_sweep.py 1500 --gpu --seed 61
run split python tools/param_sweep.py 800 --gpu --split --seed 62
run fields python tools/param_sweep.py 600 --gpu --fields --seed 63
run fields_split python tools/param_sweep.py 400 --gpu --fields --split --seed 64
run batch python tools/param_sweep.py 300 --gpu --batch 6 --seed 65
run concat python tools/param_sweep.py 300 --gpu --concat --seed 66
run extreme python tools/param_sweep.py 500 --gpu --extreme --seed 67
run xwgs python tools/param_sweep.py 400 --gpu --xwgs --seed 68
run pocd python tools/param_sweep.py 300 --gpu --pocdelta --seed 69
