#!/bin/bash
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $out/p1 -- python3 $R/"$@" > $out/p1.log 2>&1 || tail -5 $out/p1.log
python3 $R/tools/pmc_summary.py $out/p1 | grep -A9 "k_deblock\|k_inter\|k_intra"
