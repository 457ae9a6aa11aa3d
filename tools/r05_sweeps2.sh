#!/bin/bash
# Round 5, second set (seeds 601..612) on the final build: bash tools/r05_sweeps2.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
tag=$1; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; timeout -k 10 900 "$@" > $out/$name.log 2>&1; echo "$name rc=$? $(tail -1 $out/$name.log | cut -c1-160)"; }
run plain python tools/param_sweep.py 1500 --gpu --seed 601
run split python tools/param_sweep.py 800 --gpu --split --seed 602
run fields python tools/param_sweep.py 600 --gpu --fields --seed 603
run fields_split python tools/param_sweep.py 400 --gpu --fields --split --seed 604
run batch python tools/param_sweep.py 300 --gpu --batch 6 --seed 605
run concat python tools/param_sweep.py 300 --gpu --concat --seed 606
run extreme python tools/param_sweep.py 500 --gpu --extreme --seed 607
run xwgs python tools/param_sweep.py 400 --gpu --xwgs --seed 608
run pocd python tools/param_sweep.py 300 --gpu --pocdelta --seed 609
run big python tools/param_sweep.py 150 --gpu --big --seed 610
run fields_extreme python tools/param_sweep.py 300 --gpu --fields --extreme --seed 611
run fields_xwgs python tools/param_sweep.py 300 --gpu --fields --xwgs --seed 612
