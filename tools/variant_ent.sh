# A/B of entropy-kernel register budgets (MI_ENT_MINWAVES): variant libraries built with
#   make -C h264decode_amd/csrc EXTRA=-DMI_ENT_MINWAVES=7 BUILD=_build_ew7 OUT=../libh264mi_ew7.so
for lib in "" ${VARIANTS:-h264decode_amd/libh264mi_ew7.so h264decode_amd/libh264mi_ew8.so}; do H264MI_LIB=$lib timeout -k 10 300 python bench.py --steps 4 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"; done
