#!/bin/bash
# Round 5, further seed sets on the final build (7xx / 8xx)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
tag=$1; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; timeout -k 10 900 "$@" > $out/$name.log 2>&1; echo "$name rc=$? $(tail -1 $out/$name.log | cut -c1-160)"; }
run plain python tools/param_sweep.py 1500 --gpu --seed 701
run split python tools/param_sweep.py 800 --gpu --split --seed 702
run fields python tools/param_sweep.py 600 --gpu --fields --seed 703
run fields_split python tools/param_sweep.py 400 --gpu --fields --split --seed 704
run batch python tools/param_sweep.py 300 --gpu --batch 6 --seed 705
run concat python tools/param_sweep.py 300 --gpu --concat --seed 706
run extreme python tools/param_sweep.py 500 --gpu --extreme --seed 707
run xwgs python tools/param_sweep.py 400 --gpu --xwgs --seed 708
run pocd python tools/param_sweep.py 300 --gpu --pocdelta --seed 709
run big python tools/param_sweep.py 150 --gpu --big --seed 710
run fields_extreme python tools/param_sweep.py 300 --gpu --fields --extreme --seed 711
run fields_xwgs python tools/param_sweep.py 300 --gpu --fields --xwgs --seed 712
