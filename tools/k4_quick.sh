#!/bin/bash
# Quick K4 / K3 loop on the GPU box: parity of the matrix in the default kernel plan, the full-size cases, the bench line's kernel times (32 distinct streams).
#   bash tools/k4_quick.sh <tag>
set -o pipefail
tag=${1:-k4}
out=gpurun_out
mkdir -p $out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/intrin_probe.hip -o /tmp/intrin_probe 2>/dev/null && /tmp/intrin_probe | tail -3
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matches_oracle_and_generator or 1080p_full_size or 4k_high or 1080i or b_and_p_streams or cabac_field" > $out/${tag}_pytest.log 2>&1 || { tail -20 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --distinct 32 > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -5 $out/${tag}_bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$out/${tag}_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "kernels", r["all_kernels_ms_per_step"], "k_inter ms", r["per_launch"]["k_inter"]["ms"], "k_intra ms", r["per_launch"]["k_intra"]["ms"])
PY
