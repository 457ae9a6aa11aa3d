#!/usr/bin/env python3
"""Diagnostics: per-slice entropy-kernel time and bin count (needs a build with EXTRA=-DMI_ENT_STATS=1 for bins).
Usage: H264MI_SLICE_STATS=1 python tools/slice_stats.py [streams] [frames] [recipe]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("H264MI_SLICE_STATS", "1")
import streamgen
import h264decode_amd as H

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
R = sys.argv[3] if len(sys.argv) > 3 else "C3"  # recipe: C3 (1080p CABAC I P P P) or C2 (720p CAVLC all-intra)
kw = streamgen.recipe("C3", frames=F, idr_period=F, seed=1000, width=1920, height=1080) if R == "C3" else streamgen.recipe(R, frames=F, seed=1000)
s, rec, sizes = streamgen.encode(want_recon=True, **kw)
dec = H.Decoder(max_streams=S, max_width=1920, max_height=1088, max_frames_per_batch=F, max_slices_per_frame=1, coef_blocks_per_mb=16)
dec.prepare([s] * S)
dec.execute(); dec.sync()
print("second pass", file=sys.stderr)
dec.execute(); dec.sync()
