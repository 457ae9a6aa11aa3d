R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "" _dp32 _dp128; do
  out=$R/gpurun_out/r06e$v; mkdir -p $out
  H264MI_LIB=$R/h264decode_amd/libh264mi$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-parity --distinct 32 --serialized > $out/bench.json 2> $out/err.txt
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "variant '$v'"; grep "k_dbprep\|k_entropy" $f | cut -d, -f1-4
  rm -rf $out
done
