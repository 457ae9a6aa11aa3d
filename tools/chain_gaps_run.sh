R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
PROBE_X=256 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps -o gaps -- python3 $R/tools/lowstream_probe.py 1 > $R/gpurun_out/gaps_probe.log 2>&1
python3 $R/tools/chain_gaps.py $R/gpurun_out/gaps > $R/gpurun_out/gaps.txt 2>&1
cat $R/gpurun_out/gaps.txt
rm -rf $R/gpurun_out/gaps
