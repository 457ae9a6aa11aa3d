#!/bin/bash
# A/B of prebuilt variant libraries on the GPU box: bash tools/ab_libs.sh <tag> "" h264decode_amd/libh264mi_v1.so ...   ("" = the default build)
# Each variant: the GPU matrix (bit-exactness), then the headline bench line without the extra keys.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
tag=$1; shift
mkdir -p gpurun_out/$tag
for lib in "$@"; do
  name=$(basename "${lib:-default}" .so)
  H264MI_LIB=$lib timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "matrix" > gpurun_out/$tag/$name.pytest.log 2>&1 || { echo "$name: PARITY FAILED"; tail -5 gpurun_out/$tag/$name.pytest.log; continue; }
  for rep in 1 2; do
    echo -n "$name rep $rep: "
    H264MI_LIB=$lib timeout -k 10 300 python bench.py --distinct 32 --no-extra --no-cpu-baseline --no-parity --steps 5 2>gpurun_out/$tag/$name.err | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
  done
done
