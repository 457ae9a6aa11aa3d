#!/bin/bash
# Round 5, seed set 9xx on the final build
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
tag=$1; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; timeout -k 10 900 "$@" > $out/$name.log 2>&1; echo "$name rc=$? $(tail -1 $out/$name.log | cut -c1-160)"; }
run plain python tools/param_sweep.py 1500 --gpu --seed 901
run split python tools/param_sweep.py 800 --gpu --split --seed 902
run fields python tools/param_sweep.py 600 --gpu --fields --seed 903
run fields_split python tools/param_sweep.py 400 --gpu --fields --split --seed 904
run batch python tools/param_sweep.py 300 --gpu --batch 6 --seed 905
run concat python tools/param_sweep.py 300 --gpu --concat --seed 906
run extreme python tools/param_sweep.py 500 --gpu --extreme --seed 907
run xwgs python tools/param_sweep.py 400 --gpu --xwgs --seed 908
run pocd python tools/param_sweep.py 300 --gpu --pocdelta --seed 909
run big python tools/param_sweep.py 150 --gpu --big --seed 910
run fields_extreme python tools/param_sweep.py 300 --gpu --fields --extreme --seed 911
run fields_xwgs python tools/param_sweep.py 300 --gpu --fields --xwgs --seed 912
