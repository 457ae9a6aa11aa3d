#!/bin/bash
# Builds the host side of libh264mi against the null device (tools/hoststub) and tools/host_pocs.cpp into
# ${TMPDIR:-/tmp}/h264mi_host_pocs/ and prints the path of the program.  No GPU, no HIP toolchain needed (g++).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=${TMPDIR:-/tmp}/h264mi_host_pocs
mkdir -p "$out"
g++ -std=c++17 -O1 -g -fPIC -shared -I"$root/tools/hoststub" -I"$root/include" "$root"/h264decode_amd/csrc/mi_api.cpp "$root"/h264decode_amd/csrc/mi_parse.cpp \
    "$root"/h264decode_amd/csrc/mi_cabac_mn.cpp -o "$out/libh264mi_host.so" -lpthread
g++ -std=c++17 -O1 -g -I"$root/include" "$root/tools/host_pocs.cpp" -L"$out" -lh264mi_host -Wl,-rpath,"$out" -o "$out/host_pocs"
echo "$out/host_pocs"
