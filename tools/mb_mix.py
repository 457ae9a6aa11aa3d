#!/usr/bin/env python3
"""Diagnostics: macroblock mix of the bench recipe's P pictures (types, coded mb_type, residual blocks per macroblock), from the records
the entropy kernel writes.  Usage: python tools/mb_mix.py [frames]"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import streamgen
import h264decode_amd as H

F = int(sys.argv[1]) if len(sys.argv) > 1 else 6
s, rec, sizes = streamgen.encode(want_recon=True, **streamgen.recipe("C3", frames=F, idr_period=F, seed=1000, width=1920, height=1080))
dec = H.Decoder(max_streams=1, max_width=1920, max_height=1088, max_frames_per_batch=F, max_slices_per_frame=1)
dec.decode([s])
names = {0: "NONE", 1: "I4x4", 2: "I8x8", 3: "I16x16", 4: "IPCM", 5: "P16x16", 6: "P16x8", 7: "P8x16", 8: "P8x8", 9: "PSKIP"}
for f in (1, F - 1):
    r = dec.read_mbrecs(0, f, 8160)
    t = collections.Counter(int(x) for x in r[:, 0])
    cbp = r[:, 5]
    nz = np.array([bin(int(a) | int(b) << 8).count("1") for a, b in zip(r[:, 8], r[:, 9])])
    coded = r[:, 0] != 9
    print("frame", f, {names.get(k, k): v for k, v in sorted(t.items())}, "| coded MBs: cbp == 0:", int(((cbp == 0) & coded).sum()), "luma 4x4 blocks with coefficients per coded MB: %.2f" % nz[coded].mean(),
          "chroma cbp 0/1/2:", [int((((cbp >> 4) == k) & coded).sum()) for k in range(3)])
    sub = collections.Counter(int(x) for x in r[r[:, 0] == 8][:, 16 + 5:16 + 9].reshape(-1)) if (r[:, 0] == 8).any() else {}
    print("   sub_mb_type of P8x8 quadrants:", dict(sub))
    # K4's view: inter macroblocks by vector shape (record bytes 48..111: sixteen (mvx, mvy) int16 pairs)
    inter = np.isin(r[:, 0], (5, 6, 7, 8, 9))
    mv = r[:, 48:112].copy().view(np.int16).reshape(-1, 16, 2)
    uni = (mv == mv[:, :1, :]).all(axis=(1, 2))
    integer = ((mv & 3) == 0).all(axis=(1, 2))
    cbp0 = cbp == 0
    n = int(inter.sum())
    print("   inter MBs %d: one vector %.1f %%, integer vectors %.1f %%, both %.1f %%, both and no residual %.1f %%, no residual %.1f %%, 8x8 transform %.1f %%" % (
        n, 100.0 * (uni & inter).sum() / n, 100.0 * (integer & inter).sum() / n, 100.0 * (uni & integer & inter).sum() / n, 100.0 * (uni & integer & inter & cbp0).sum() / n,
        100.0 * (inter & cbp0).sum() / n, 100.0 * (inter & (r[:, 1] != 0)).sum() / n))
