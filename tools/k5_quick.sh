#!/bin/bash
# Quick K5 loop on the GPU box: parity of the one-workgroup-per-picture kernels on the whole matrix, the bench line's kernel times (32 distinct streams:
# short input generation), and the phase clocks of k_deblock from the -DMI_DB_STATS build if it is there.   bash tools/k5_quick.sh <tag>
set -o pipefail
tag=${1:-k5}
out=gpurun_out
mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_workgroup_per_picture or 1080p_full_size or 4k_high or 1080i" > $out/${tag}_pytest.log 2>&1 || { tail -20 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --distinct 32 > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -5 $out/${tag}_bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$out/${tag}_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "kernels", r["all_kernels_ms_per_step"], "k_deblock ms", r["per_launch"]["k_deblock"]["ms"], "k_inter ms", r["per_launch"]["k_inter"]["ms"])
PY
if [ -f h264decode_amd/libh264mi_stats.so ]; then
  H264MI_LIB=h264decode_amd/libh264mi_stats.so timeout -k 10 300 python tools/deblock_phase_probe.py 256 4 > $out/${tag}_phases.txt 2>&1; cat $out/${tag}_phases.txt
fi
for nw in ${K5_WAVES_AB:-}; do
  H264MI_K5_WAVES=$nw timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --distinct 32 --no-parity > $out/${tag}_bench_w$nw.json 2> $out/${tag}_bench_w$nw.err
  python - <<PY
import json
d = json.loads(open("$out/${tag}_bench_w$nw.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("K5_WAVES=$nw value", d["value"], "kernels", r["all_kernels_ms_per_step"], "k_deblock ms", r["per_launch"]["k_deblock"]["ms"])
PY
done
