#!/usr/bin/env python3
"""Call-sequence fuzz of the C ABI (not a test): random interleavings of prepare / execute / sync / decode / reset / stream reset /
isolation / frame reads / pack / status on one decoder, with good, empty, truncated and foreign-size inputs.  Every call must come back
with a status code -- never a crash or a hang -- and a good decode after any sequence must still be bit-exact.  Usage: api_fuzz.py [rounds]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import streamgen
import h264decode_amd as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(3)
good = [streamgen.encode(width=176, height=144, frames=5, idr_period=0, profile_idc=77, cabac=1, seed=1),
        streamgen.encode(width=96, height=80, frames=4, idr_period=2, profile_idc=66, cabac=0, slice_groups=3, fmo_type=1, aso=1, seed=2),
        streamgen.encode(width=176, height=144, frames=6, idr_period=0, profile_idc=77, cabac=1, bframes=2, num_ref_frames=3, seed=3),
        # field pictures (a frame = two pictures; cut anywhere, the first field of a frame may be left waiting for its second one)
        streamgen.encode(width=176, height=128, frames=4, idr_period=0, profile_idc=77, cabac=0, field_pics=3, num_ref_frames=2, mmco=1, seed=5)]
NG = len(good)
DIMS = [(176, 144), (96, 80), (176, 144), (176, 128)]
big = streamgen.encode(width=352, height=288, frames=2, idr_period=0, profile_idc=77, cabac=1, seed=4)[0]  # larger than the decoder allows
codes = {}
t0 = time.time()
for r in range(N):
    dec = H.Decoder(max_streams=2, max_width=176, max_height=144, max_frames_per_batch=8, max_slices_per_frame=8)
    L, h = dec._L, dec._h
    log = []
    for step in range(int(rng.integers(3, 25))):
        op = int(rng.integers(0, 14))
        state = rng.bit_generator.state["state"]["state"]  # (to replay the draws of this step)
        log.append((op, state))
        try:
            if op == 0:
                dec.prepare([good[int(rng.integers(0, NG))][0], good[int(rng.integers(0, NG))][0]])
            elif op == 1:
                dec.execute()
            elif op == 2:
                dec.sync()
            elif op == 3:
                s = good[int(rng.integers(0, NG))][0]
                dec.decode([s[:int(rng.integers(0, len(s)))], b""])
            elif op == 4:
                dec.reset()
            elif op == 5:
                dec.reset_stream(int(rng.integers(-1, 3)))
            elif op == 6:
                dec.set_isolation(bool(rng.integers(0, 2)))
            elif op == 7:
                dec.read_frame(int(rng.integers(-1, 3)), int(rng.integers(-1, 8)), bool(rng.integers(0, 2)))
            elif op == 8:
                dec.stream_status(int(rng.integers(-1, 3)))
            elif op == 9:
                dec.decode([big, good[0][0]])
            elif op == 10:
                dec.frame_info(int(rng.integers(-1, 3)), int(rng.integers(-1, 8)))
            elif op == 11:
                dec.output_order(int(rng.integers(-1, 3)))
            elif op == 12:
                dec.read_mbrecs(int(rng.integers(0, 2)), int(rng.integers(0, 7)), 99)
            else:
                dec.decode([good[0][0], good[2][0]])
            codes[0] = codes.get(0, 0) + 1
        except H.H264MIError as e:
            codes[e.code] = codes.get(e.code, 0) + 1
            assert e.code in (-1, -2, -3, -7, -8), e
        except (IndexError, ValueError):
            codes["py"] = codes.get("py", 0) + 1
    # whatever happened: a clean decode afterwards is exact
    k = int(rng.integers(0, NG))
    try:
        dec.reset()
        dec.set_isolation(False)
        dec.decode([good[k][0], good[(k + 1) % NG][0]])
        for i, g in ((0, good[k]), (1, good[(k + 1) % NG])):
            w, hh = DIMS[(k + i) % NG]
            assert np.array_equal(dec.read_frames(i, crop=False, size=w * hh * 3 // 2), g[1]), (r, i)
    except Exception as e:  # noqa: BLE001
        print("FAILED round %d k=%d after ops %s: %s" % (r, k, [o for o, _ in log], repr(e)[:200]), flush=True)
        raise
    dec.close()
    if (r + 1) % 50 == 0:
        print("round %d codes %s %.1fs" % (r + 1, codes, time.time() - t0), flush=True)
print("api fuzz OK", codes)
