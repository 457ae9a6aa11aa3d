#!/usr/bin/env python3
"""Per-launch times of K4 / K3 / K5 at low stream counts, with the banded kernels on and off.
Usage: lowstream_probe.py [streams ...]   (default: 1 32)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import streamgen
import h264decode_amd as H

F = 30
nd = 4
mot = [int(v) for v in os.environ.get("PROBE_MOTION", "12,-8").split(",")]  # scene motion per frame in quarter samples
gen = [streamgen.encode(want_recon=True, **streamgen.recipe("C3", frames=F, idr_period=F, seed=1000 + i, width=1920, height=1080, motion_x4=mot[0], motion_y4=mot[1])) for i in range(nd)]
for S in [int(a) for a in sys.argv[1:]] or [1, 32]:
    gops = 10 if S == 1 else 1
    streams = [b"".join(gen[(i + j) % nd][0] for j in range(gops)) for i in range(S)]
    for x in os.environ.get("PROBE_X", "0,256").split(","):
        os.environ["H264MI_X_WGS"] = x
        dec = H.Decoder(max_streams=S, max_width=1920, max_height=1088, max_frames_per_batch=F * gops, max_slices_per_frame=1)
        dec.prepare(streams)
        dec.execute(); dec.sync()
        ok = np.array_equal(dec.read_frames(0, crop=False)[:F], gen[0][1])
        dec.set_profiling(True)
        dec.execute(); dec.sync()
        kt = dec.kernel_times_ms()
        lt = {k: np.array(dec.launch_times_ms(k)) for k in ("inter", "intra", "deblock")}
        dec.set_profiling(False)
        t0 = time.perf_counter()
        for _ in range(3):
            dec.execute()
        dec.sync()
        dt = (time.perf_counter() - t0) / 3
        print("S=%d x_wgs=%s parity=%s kernel ms %s | per launch: inter %.3f intra I %.3f P %.3f deblock %.3f | pipelined %.1f ms/pass = %.0f fps" % (
            S, x, ok, {k: round(v, 1) for k, v in kt.items()}, lt["inter"].mean(), lt["intra"][0], lt["intra"][1:].mean(), lt["deblock"].mean(), dt * 1e3,
            S * F * gops / dt), flush=True)
        dec.close()
