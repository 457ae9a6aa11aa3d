#!/bin/bash
# Round-5 evidence in one GPU-box call: the default bench line, kernel-trace stats of the bench command, HBM traffic (two --pmc passes) and the
# instruction mix (two --pmc passes).  Counter passes never share a run with tracing domains other than --kernel-trace.
# Usage (GPU box, repo root): bash tools/r05_profile.sh <tag>
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout -k 10 800 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"; tail -c 300 $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o $tag -- python3 $R/bench.py --steps 3 --no-cpu-baseline --no-extra > $out/${tag}_stats_bench.json 2> $out/stats.err; echo "stats rc=$?"
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
echo "--- kernel stats"; head -12 $out/${tag}_kernel_stats.csv
# the same with the passes serialized (library profiling mode: one stream, stages back to back): the averages of THIS table are the quantity the bench line's launch_ms is
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_ser -o $tag -- python3 $R/bench.py --steps 3 --no-cpu-baseline --no-extra --no-parity --serialized > $out/${tag}_stats_serialized_bench.json 2> $out/stats_ser.err; echo "serialized stats rc=$?"
find $out/stats_ser -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats_serialized.csv \;
echo "--- kernel stats, serialized"; head -8 $out/${tag}_kernel_stats_serialized.csv
cd $R && bash tools/pmc_traffic.sh ${tag}_traffic --no-extra > $out/${tag}_pmc_traffic.txt 2>&1; echo "traffic rc=$?"; tail -9 $out/${tag}_pmc_traffic.txt
cd $R && bash tools/pmc_run.sh ${tag}_mix bench.py --no-extra --no-cpu-baseline --no-parity --steps 2 > $out/${tag}_pmc_instruction_mix.txt 2>&1; echo "mix rc=$?"; grep -A8 "^k_entropy" $out/${tag}_pmc_instruction_mix.txt | head -24
cp $R/profiles/pmc_traffic.json $out/pmc_traffic.json
rm -rf $out/stats $out/stats_ser $R/gpurun_out/${tag}_traffic/FETCH_SIZE $R/gpurun_out/${tag}_traffic/WRITE_SIZE $R/gpurun_out/${tag}_mix/p1 $R/gpurun_out/${tag}_mix/p2
