import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, streamgen, h264decode_amd as H
W, Hh = 64, 48
for my in range(4):
    for mx in range(4):
        kw = dict(width=W, height=Hh, frames=2, idr_period=0, profile_idc=66, cabac=0, deblock_idc=1, motion_x4=8 + mx, motion_y4=-4 + my, qp=40, skip_permille=0, intra_in_p_permille=0, sub8x8_permille=0, noise=4)
        s, rec, _ = streamgen.encode(**kw)
        dec = H.Decoder(max_streams=1, max_width=W, max_height=Hh, max_frames_per_batch=2)
        dec.decode([s])
        out = dec.read_frames(0, crop=False, size=W * Hh * 3 // 2)
        recs = dec.read_mbrecs(0, 1, 12)
        mv = recs[:, 48:52].copy().view(np.int16).reshape(-1, 2)
        Yg, Yr = out[1][:W * Hh].reshape(Hh, W).astype(int), rec[1][:W * Hh].reshape(Hh, W).astype(int)
        badmb = (Yg != Yr).reshape(3, 16, 4, 16).any(axis=(1, 3)).reshape(-1)
        cls_bad = sorted({(int(mv[m, 0]) & 3, int(mv[m, 1]) & 3) for m in range(12) if badmb[m] and recs[m, 0] >= 5})
        cls_ok = sorted({(int(mv[m, 0]) & 3, int(mv[m, 1]) & 3) for m in range(12) if not badmb[m] and recs[m, 0] >= 5})
        print("scene (%d,%d): bad classes %s ok classes %s" % (mx, my, cls_bad, cls_ok))
        if cls_bad and mx + my * 4 in (1, 4, 5):
            m = [m for m in range(12) if badmb[m]][0]
            print("  mb", m, "mv", mv[m].tolist(), "cmask %#x" % int(recs[m, 116:120].copy().view(np.uint32)[0]))
            x0, y0 = (m % 4) * 16, (m // 4) * 16
            for r in range(16):
                print("   " + " ".join("%4d" % v for v in (Yg - Yr)[y0 + r, x0:x0 + 16]))
        dec.close()
