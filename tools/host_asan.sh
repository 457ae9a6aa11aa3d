#!/bin/bash
# The host side of libh264mi (mi_api.cpp + mi_parse.cpp: parsing, picture boundaries, DPB / reference lists, slice-group maps,
# batching, staging layout, error paths) under AddressSanitizer + UBSan + LeakSanitizer on the CPU, against a null device
# (tools/hoststub: kernels are not run), fed intact and damaged copies of the whole test matrix.  No GPU needed.
#   bash tools/host_asan.sh [decoders] [seed]          (SAN=thread for ThreadSanitizer: the per-stream prepare threads;
#                                                       H264MI_HOST_THREADS=N: N clients side by side, each with its own decoders)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=${TMPDIR:-/tmp}/h264mi_host_asan
mkdir -p "$out"
san="-std=c++17 -O1 -g -fsanitize=${SAN:-address,undefined} -fno-omit-frame-pointer"
g++ $san -fPIC -shared -I"$root/tools/hoststub" -I"$root/include" "$root"/h264decode_amd/csrc/mi_api.cpp "$root"/h264decode_amd/csrc/mi_parse.cpp \
    "$root"/h264decode_amd/csrc/mi_cabac_mn.cpp -o "$out/libh264mi_hostasan.so" -lpthread
g++ $san -I"$root/include" "$root/tools/host_asan.cpp" -L"$out" -lh264mi_hostasan -Wl,-rpath,"$out" -o "$out/host_asan"
python3 - "$out" <<PY
import sys
sys.path.insert(0, "$root"); sys.path.insert(0, "$root/tests")
import streamgen
from conftest import FIELD_MATRIX, MATRIX
for n, k in list(MATRIX.items()) + list(FIELD_MATRIX.items()):  # (field pictures: the product must refuse them cleanly)
    open("%s/m_%s.h264" % (sys.argv[1], n), "wb").write(streamgen.encode(**k)[0])
PY
"$out/host_asan" "${1:-300}" "${2:-1}" "$out"/m_*.h264
