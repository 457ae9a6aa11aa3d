#!/bin/bash
# Round 5: every sweep mode with fresh seeds on the new K5, CABAC field pictures (allow_unpinned_field_cabac) and monochrome recipes: bash tools/r05_sweeps.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
tag=$1; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; timeout -k 10 900 "$@" > $out/$name.log 2>&1; echo "$name rc=$? $(tail -1 $out/$name.log | cut -c1-160)"; }
run plain python tools/param_sweep.py 1500 --gpu --seed 501
run split python tools/param_sweep.py 800 --gpu --split --seed 502
run fields python tools/param_sweep.py 600 --gpu --fields --seed 503
run fields_split python tools/param_sweep.py 400 --gpu --fields --split --seed 504
run batch python tools/param_sweep.py 300 --gpu --batch 6 --seed 505
run concat python tools/param_sweep.py 300 --gpu --concat --seed 506
run extreme python tools/param_sweep.py 500 --gpu --extreme --seed 507
run xwgs python tools/param_sweep.py 400 --gpu --xwgs --seed 508
run pocd python tools/param_sweep.py 300 --gpu --pocdelta --seed 509
run big python tools/param_sweep.py 150 --gpu --big --seed 510
run fields_extreme python tools/param_sweep.py 300 --gpu --fields --extreme --seed 511
run fields_xwgs python tools/param_sweep.py 300 --gpu --fields --xwgs --seed 512
