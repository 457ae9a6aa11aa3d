#!/usr/bin/env python3
"""Where a K5 step spends its time: shader clocks per phase of the wavefront of group 0 of every picture, from a -DMI_DB_STATS build.
Usage: deblock_phase_probe.py [streams] [frames]   (1 stream: the banded kernel k_deblock_x on one picture; 256 streams: k_deblock proper,
one workgroup per picture, every CU busy -- the launch shape of the bench).  Build first (no GPU needed):
  make -C h264decode_amd/csrc EXTRA="-DMI_DB_STATS -DH264MI_TEST_HOOKS" BUILD=_build_stats OUT=../libh264mi_stats.so
then on the GPU box: H264MI_LIB=h264decode_amd/libh264mi_stats.so python tools/deblock_phase_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import streamgen
import h264decode_amd as H

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
s, rec, _ = streamgen.encode(want_recon=True, **streamgen.recipe("C3", frames=F, idr_period=F, seed=1000, width=1920, height=1080))
dec = H.Decoder(max_streams=S, max_width=1920, max_height=1088, max_frames_per_batch=F, max_slices_per_frame=1)
f = H.load().h264mi_internal_deblock_phase_clocks  # (H264MI_LIB = the -DMI_DB_STATS -DH264MI_TEST_HOOKS build)
f.restype, f.argtypes = ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
buf = (ctypes.c_uint32 * 12)()
dec.decode([s] * S)
assert np.array_equal(dec.read_frames(S - 1, crop=False), rec)
f(dec._h, buf)
dec.decode([s] * S)
f(dec._h, buf)
if S > 1:  # k_deblock proper (round 5: groups of 8 rows, 129 steps)
    names = ["0a inputs out of the prefetch registers", "1 vertical edges", "2 hand-off", "3 horizontal edges", "0b output of column x - 2", "loop", "4c DbPrm stage + sync", "0c prefetch issue", "4a wait for the loads", "4b pieces into the window", "-", "-"]
    nst = 129
else:      # the banded kernel k_deblock_x
    names = ["1c parameters (rest of 1)", "2 vertical edges", "3 hand-off", "4 horizontal edges", "5 results", "loop", "1a DbPrm -> LDS", "1b prefetch issue", "-", "-", "-", "-"]
    nst = 123
steps = S * F * float(nst)
tot = sum(buf)
print("clocks per step of the wavefront of group 0 (1080p, %d streams x %d pictures, %d steps each):" % (S, F, nst))
for n, v in zip(names, buf):
    print("  %-20s %8.0f  %5.1f %%" % (n, v / steps, 100.0 * v / tot))
print("  %-20s %8.0f" % ("total", tot / steps))
if S > 1 and hasattr(H.load(), "h264mi_internal_deblock_group_times"):
    g = H.load().h264mi_internal_deblock_group_times
    g.restype, g.argtypes = ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
    tb = (ctypes.c_uint32 * 128)()
    g(dec._h, tb)
    t0 = min(tb[2 * k] for k in range(9))
    print("row groups of the first picture of the last launch, start .. end in us:", ", ".join("%.0f..%.0f" % ((tb[2 * k] - t0) * 0.01, (tb[2 * k + 1] - t0) * 0.01) for k in range(9)))
    for k in range(9):
        print("  group %d begins step 0, 16, 32, ... at us:" % k, " ".join("%5.0f" % ((tb[32 + 8 * k + i] - t0) * 0.01) for i in range(8)))
