#!/bin/bash
# Usage: bash tools/pmc_run.sh <tag> <python script + args>   -- two counter passes (instruction mix, cycles)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $out/p1 -- python3 $R/"$@" > $out/p1.log 2>&1 || tail -5 $out/p1.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d $out/p2 -- python3 $R/"$@" > $out/p2.log 2>&1 || tail -5 $out/p2.log
python3 $R/tools/pmc_summary.py $out/p1 $out/p2
