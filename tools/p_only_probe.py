#!/usr/bin/env python3
"""Experiment: entropy / reconstruction time of a batch WITHOUT I slices (frames 1..29 of every GOP; the I frames are decoded by
a batch of their own first).  Answers: how long is a pass if the I-slice latency is hidden?  Usage: p_only_probe.py [streams]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import streamgen
import h264decode_amd as H

S = int(sys.argv[1]) if len(sys.argv) > 1 else 204
F = 30
nd = 4
gen = [streamgen.encode(want_recon=True, **streamgen.recipe("C3", frames=F, idr_period=F, seed=1000 + i, width=1920, height=1080)) for i in range(nd)]
streams = [gen[i % nd] for i in range(S)]
dec = H.Decoder(max_streams=S, max_width=1920, max_height=1088, max_frames_per_batch=F, max_slices_per_frame=1)
heads = [s[0][:int(s[2][0])] for s in streams]
tails = [s[0][int(s[2][0]):] for s in streams]
for label, batch in (("I only", heads), ("P only", tails)):
    dec.prepare(batch)
    dec.execute(); dec.sync()
    dec.set_profiling(True)
    dec.execute(); dec.sync()
    kt = dec.kernel_times_ms()
    dec.set_profiling(False)
    t0 = time.perf_counter()
    for _ in range(4):
        dec.execute()
    dec.sync()
    print(label, "streams", S, "frames", dec.frame_count(0), "kernel ms", {k: round(v, 1) for k, v in kt.items()}, "pipelined ms/pass %.1f" % ((time.perf_counter() - t0) / 4 * 1e3))
got = dec.read_frames(S - 1, crop=False)
print("parity of the P batch:", np.array_equal(got, streams[S - 1][1][1:]))
