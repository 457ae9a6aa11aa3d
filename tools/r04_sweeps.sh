#!/bin/bash
# Randomised sweeps and fuzzers on the GPU box after a scheduling change: bash tools/r04_sweeps.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
tag=$1; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; timeout -k 10 600 "$@" > $out/$name.log 2>&1; echo "$name rc=$? $(tail -1 $out/$name.log | cut -c1-160)"; }
run sweep python tools/param_sweep.py 300 --gpu --seed 41
run sweep_split python tools/param_sweep.py 300 --gpu --split --seed 42
run sweep_fields python tools/param_sweep.py 200 --gpu --fields --seed 43
run sweep_fields_split python tools/param_sweep.py 200 --gpu --fields --split --seed 44
run sweep_batch python tools/param_sweep.py 150 --gpu --batch 6 --seed 45
run sweep_concat python tools/param_sweep.py 150 --gpu --concat --seed 46
run sweep_pocd python tools/param_sweep.py 150 --gpu --pocdelta --seed 47
run sweep_xwgs python tools/param_sweep.py 200 --gpu --xwgs --seed 48
run fuzz python tools/fuzz_gpu.py
run api_fuzz python tools/api_fuzz.py
run server_fuzz python tools/server_fuzz.py
