#!/usr/bin/env python3
"""Diagnostics: kernel times and per-slice entropy times of a B-picture batch (I B B P, 1080p).
Usage: python tools/b_profile.py [streams] [frames] [key=value generator overrides ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("H264MI_SLICE_STATS", "1")
if os.environ["H264MI_SLICE_STATS"] == "0":
    del os.environ["H264MI_SLICE_STATS"]
import numpy as np
import streamgen
import h264decode_amd as H

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
F = int(sys.argv[2]) if len(sys.argv) > 2 else 30
over = dict(bframes=2, num_ref_frames=3, bskip_permille=300)
for a in sys.argv[3:]:
    k, v = a.split("=")
    over[k] = int(v)
kw = dict(streamgen.recipe("C3", frames=F, idr_period=F, seed=3000, width=1920, height=1080), **over)
s, rec, sizes = streamgen.encode(want_recon=True, **kw)
print("stream bytes", len(s), "per picture", [int(x) for x in sizes[:8]], file=sys.stderr)
dec = H.Decoder(max_streams=S, max_width=1920, max_height=1088, max_frames_per_batch=F, max_slices_per_frame=1)
dec.prepare([s] * S)
dec.execute(); dec.sync()
assert np.array_equal(dec.read_frames(S - 1, crop=False), rec)
dec._L.h264mi_decoder_set_profiling(dec._h, 1)
dec.execute(); dec.sync()
print("kernel ms (entropy, inter, intra, deblock, total):", dec.kernel_times_ms() if hasattr(dec, "kernel_times_ms") else None, file=sys.stderr)
dec._L.h264mi_decoder_set_profiling(dec._h, 0)
t0 = time.perf_counter()
for _ in range(6):
    dec.execute()
dec.sync()
print("ms per pass (pipelined):", (time.perf_counter() - t0) / 6 * 1e3, file=sys.stderr)
