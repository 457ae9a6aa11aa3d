#!/bin/bash
# Round 5, further seed sets on the final build (7xx / 8xx)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
tag=$1; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; timeout -k 10 900 "$@" > $out/$name.log 2>&1; echo "$name rc=$? $(tail -1 $out/$name.log | cut -c1-160)"; }
run plain python tools/param_sweep.py 1500 --gpu --seed 801
run split python tools/param_sweep.py 800 --gpu --split --seed 802
run fields python tools/param_sweep.py 600 --gpu --fields --seed 803
run fields_split python tools/param_sweep.py 400 --gpu --fields --split --seed 804
run batch python tools/param_sweep.py 300 --gpu --batch 6 --seed 805
run concat python tools/param_sweep.py 300 --gpu --concat --seed 806
run extreme python tools/param_sweep.py 500 --gpu --extreme --seed 807
run xwgs python tools/param_sweep.py 400 --gpu --xwgs --seed 808
run pocd python tools/param_sweep.py 300 --gpu --pocdelta --seed 809
run big python tools/param_sweep.py 150 --gpu --big --seed 810
run fields_extreme python tools/param_sweep.py 300 --gpu --fields --extreme --seed 811
run fields_xwgs python tools/param_sweep.py 300 --gpu --fields --xwgs --seed 812
