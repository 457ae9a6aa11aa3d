# A/B of K4 register budgets (MI_K4_WAVES wavefronts per SIMD; the default build takes what the code needs: 84 VGPRs = 5).  Build the variants first (no GPU needed):
#   for w in 6 8; do make -C h264decode_amd/csrc EXTRA=-DMI_K4_WAVES=$w BUILD=_build_k4w$w OUT=../libh264mi_k4w$w.so; done
# then on the GPU box: VARIANTS="h264decode_amd/libh264mi_k4w6.so h264decode_amd/libh264mi_k4w8.so" bash tools/variant_k4.sh
for lib in "" $VARIANTS; do H264MI_LIB=$lib timeout -k 10 300 python bench.py --steps 3 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-default}', d['value'], d['roofline']['per_launch']['k_inter'], d['roofline']['all_kernels_ms_per_step'])"; done
