#!/bin/bash
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $out/p1 -- python3 $R/"$@" > $out/p1.log 2>&1 || tail -5 $out/p1.log
python3 $R/tools/pmc_summary.py $out/p1 | grep -A8 "k_entropy"
