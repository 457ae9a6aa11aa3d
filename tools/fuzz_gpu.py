#!/usr/bin/env python3
"""Robustness fuzz (not a test): bit flips, truncations and spliced streams through the GPU decoder; every trial must end with a
status code (never a hang, never a fault).  Usage: fuzz_gpu.py [trials]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import streamgen
import h264decode_amd as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(1)
cfgs = [dict(width=176, height=144, frames=9, idr_period=0, profile_idc=77, cabac=1, bframes=2, num_ref_frames=3, direct_temporal=1, weighted_bipred=2, seed=5),
        dict(width=176, height=144, frames=9, idr_period=0, profile_idc=77, cabac=0, bframes=3, b_pyramid=1, sub8x8_permille=400, seed=6),
        dict(width=176, height=144, frames=8, idr_period=4, profile_idc=100, cabac=1, transform8x8=1, bframes=1, num_ref_frames=2, slices=3, weighted_bipred=1, seed=7),
        dict(width=176, height=144, frames=6, idr_period=0, profile_idc=77, cabac=1, num_ref_frames=3, rplm=1, mmco=1, seed=8),
        dict(width=176, height=144, frames=6, idr_period=0, profile_idc=66, cabac=0, slice_groups=4, fmo_type=6, slices=2, aso=1, seed=9),
        dict(width=176, height=144, frames=6, idr_period=0, profile_idc=66, cabac=0, slice_groups=2, fmo_type=3, aso=1, fn_gap_period=3, num_ref_frames=3, seed=10),
        dict(width=176, height=144, frames=6, idr_period=0, profile_idc=77, cabac=1, slice_groups=3, fmo_type=1, interlace_sps=0, seed=11),
        # field pictures (PAFF): all-field P and B streams, picture-adaptive with marking operations and list modification, slice groups in fields
        dict(width=176, height=128, frames=5, idr_period=0, profile_idc=77, cabac=0, field_pics=1, num_ref_frames=3, sub8x8_permille=300, seed=12),
        dict(width=176, height=128, frames=7, idr_period=0, profile_idc=77, cabac=0, field_pics=2, bframes=2, direct_temporal=1, weighted_bipred=2, num_ref_frames=2, seed=13),
        dict(width=176, height=128, frames=6, idr_period=4, profile_idc=100, cabac=0, transform8x8=1, field_pics=3, mmco=1, rplm=1, num_ref_frames=3, seed=14),
        dict(width=176, height=128, frames=4, idr_period=0, profile_idc=77, cabac=0, field_pics=3, slice_groups=2, fmo_type=3, slices=2, aso=1, num_ref_frames=2, seed=15)]
streams = [streamgen.encode(**c)[0] for c in cfgs]
codes = {}
t0 = time.time()
for t in range(N):
    s = bytearray(streams[t % len(streams)])
    kind = t % 3
    if kind == 0:
        for _ in range(int(rng.integers(1, 12))):
            i = int(rng.integers(40, len(s)))
            s[i] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        s = s[:int(rng.integers(60, len(s)))]
    else:  # splice the tail of another stream behind a cut
        o = streams[(t + 1) % len(streams)]
        s = s[:int(rng.integers(200, len(s)))] + o[int(rng.integers(100, len(o))):]
    dec = H.Decoder(max_streams=2, max_width=176, max_height=144, max_frames_per_batch=16, max_slices_per_frame=8, b_pictures=t & 2)
    dec.set_isolation(bool(t & 1))
    code = 0
    try:
        dec.decode([bytes(s), streams[0]])  # the second stream is intact
        if t & 1:
            code = dec.stream_status(0)
            assert dec.stream_status(1) == 0 and dec.frame_count(1) == cfgs[0]["frames"]
    except H.H264MIError as e:
        code = e.code
    codes[code] = codes.get(code, 0) + 1
    dec.close()
    if t % 10 == 9:
        print("trial", t + 1, "codes so far", codes, "%.1fs" % (time.time() - t0), flush=True)
assert all(c in (0, -2, -3, -7, -8) for c in codes), codes
print("fuzz OK", codes)
