for lib in "" h264decode_amd/libh264mi_k4w6.so; do H264MI_LIB=$lib timeout -k 10 300 python bench.py --steps 3 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['roofline']['per_launch']['k_inter'], d['roofline']['all_kernels_ms_per_step'])"; done
