#!/usr/bin/env python3
"""Experiment: K4 cost by macroblock class.  Decodes 64 copies of one 1080p stream whose P macroblocks are (a) all P_Skip,
(b) the bench recipe; run under rocprofv3 --pmc to read instructions per k_inter wavefront.  Usage: k4_mix_probe.py a|b"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import streamgen
import h264decode_amd as H
kw = streamgen.recipe("C3", frames=6, idr_period=6, seed=1000, width=1920, height=1080)
if sys.argv[1] == "a":
    kw.update(skip_permille=1000, intra_in_p_permille=0)
s, rec, sizes = streamgen.encode(want_recon=True, **kw)
print("bytes per P frame", int(sizes[1:].mean()))
dec = H.Decoder(max_streams=64, max_width=1920, max_height=1088, max_frames_per_batch=6)
dec.decode([s] * 64)
dec.set_profiling(True); dec.execute(); dec.sync()
print(sys.argv[1], dec.kernel_times_ms(), "inter per launch", sum(dec.launch_times_ms("inter")) / 5)
