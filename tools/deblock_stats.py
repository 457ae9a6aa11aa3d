#!/usr/bin/env python3
"""Diagnostics: per-phase shader clocks of k_deblock wavefronts (needs EXTRA=-DMI_DB_STATS=1).
[0] commit + waiting for the group above  [1] LDS sync, prefetch issue, boundary strengths  [2] the two filter passes  [3] stores, rings, progress"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import streamgen
import h264decode_amd as H
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kw = streamgen.recipe("C3", frames=3, idr_period=3, seed=1000, width=1920, height=1080)
s = streamgen.encode(**kw)[0]
dec = H.Decoder(max_streams=S, max_width=1920, max_height=1088, max_frames_per_batch=3, max_slices_per_frame=1)
dec.decode([s] * S)
for f in (0, 2):
    r = dec.read_mbrecs(0, f, 8160)[:16]
    acc = r[:, 112:128].copy().view(np.uint32).reshape(16, 4)[:9] * 16e-6
    print("frame %d: per wavefront Mclk [wait+commit, sync+bS, filters, stores]" % f)
    print(np.round(acc, 2))
    print("  sum over phases per wave:", np.round(acc.sum(1), 2), " kernel ~", round(acc.sum(1).max() / 2.39, 3), "ms")
