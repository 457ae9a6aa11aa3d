#!/usr/bin/env python3
"""Diagnostics: BASELINE configs[1] alone -- 32 streams x 60 all-IDR 720p Baseline CAVLC frames -- kernel times and pipelined rate.
Usage: python tools/c2_probe.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import streamgen
import h264decode_amd as H

fr = int(sys.argv[1]) if len(sys.argv) > 1 else 60
gen = [streamgen.encode(want_recon=True, **streamgen.recipe("C2", frames=fr, idr_period=1, seed=4000 + i)) for i in range(2)]
n = 32
cs = [gen[i % 2][0] for i in range(n)]
dec = H.Decoder(max_streams=n, max_width=1280, max_height=720, max_frames_per_batch=fr, max_slices_per_frame=1, coef_blocks_per_mb=16)
dec.prepare(cs)
dec.execute(); dec.sync()
ok = np.array_equal(dec.read_frames(n - 1, crop=False), gen[(n - 1) % 2][1])
dec.set_profiling(True)
dec.execute(); dec.sync()
kt = dec.kernel_times_ms()
dec.set_profiling(False)
t0 = time.perf_counter()
for _ in range(4):
    dec.execute()
dec.sync()
dt = (time.perf_counter() - t0) / 4
print("C2 %d streams x %d frames: parity %s, kernel ms %s, pipelined %.1f ms/pass = %.0f frames/s" % (n, fr, ok, {k: round(v, 1) for k, v in kt.items()}, dt * 1e3, n * fr / dt))
