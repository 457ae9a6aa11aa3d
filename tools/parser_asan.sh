#!/bin/bash
# Host parsers under AddressSanitizer + UBSan on the CPU (mi_parse.cpp has no HIP dependency):
# mutated SPS / PPS / slice headers and stream windows through parse_sps, parse_pps_ids,
# parse_slice_header, the slice-group maps, annexb_scan and nal_parse.
#   bash tools/parser_asan.sh [parser iterations] [seed] [oracle iterations per stream]
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=${TMPDIR:-/tmp}/h264mi_parser_asan
mkdir -p "$out"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -I"$root/include" -I"$root/h264decode_amd/csrc" "$root/tools/parser_asan.cpp" "$root/h264decode_amd/csrc/mi_parse.cpp" -o "$out/parser_asan"
python3 - "$out" <<PY
import sys
sys.path.insert(0, "$root"); sys.path.insert(0, "$root/tests")
import streamgen
from conftest import FIELD_MATRIX, MATRIX
ALL = dict(MATRIX, **FIELD_MATRIX)
for n in ("fmo_explicit", "fmo_boxout", "b_wp_explicit", "fn_gaps_cabac", "cabac_IPP", "field_b_temporal", "field_rplm_mixed_nonref"):
    if n in ALL:
        open("%s/%s.h264" % (sys.argv[1], n), "wb").write(streamgen.encode(**ALL[n])[0])
PY
for f in "$out"/*.h264; do "$out/parser_asan" "$f" "${1:-20000}" "${2:-1}"; done

# The oracle (plain C) under the same sanitizers: every stream undamaged once, then damaged copies.
gcc -std=c99 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I"$root/oracle" \
    "$root/tools/oracle_asan.c" "$root"/oracle/h264o_*.c -o "$out/oracle_asan"
for f in "$out"/*.h264; do ASAN_OPTIONS=detect_leaks=0 "$out/oracle_asan" "$f" "${3:-40}" "${2:-1}"; done
