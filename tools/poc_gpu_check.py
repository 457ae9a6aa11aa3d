"""GPU check of the POC_MATRIX streams (separate bottom-field picture order counts): product == generator reconstruction, and the PicOrderCnt the decoder reports."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
t0 = time.time()
import numpy as np
import streamgen
import h264decode_amd as H
from conftest import POC_MATRIX
bad = 0
for name, kw in sorted(POC_MATRIX.items()):
    s, rec, _ = streamgen.encode(**kw)
    W, Hc = (kw["width"] + 15) & ~15, (kw["height"] + 15) & ~15
    dec = H.Decoder(max_streams=1, max_width=W, max_height=Hc, max_frames_per_batch=kw["frames"], max_slices_per_frame=8)
    dec.decode([s])
    out = dec.read_frames(0, crop=False)
    ok = out.shape == rec.shape and np.array_equal(out, rec)
    pocs = [dec.frame_info(0, i).pic_order_cnt for i in range(kw["frames"])] if hasattr(dec, "frame_info") else None
    print(name, "OK" if ok else "MISMATCH", pocs == [int(x) for x in streamgen.last_pocs()] if pocs is not None else "", flush=True)
    bad += not ok
    dec.close()
print("poc gpu check: bad", bad, "%.1fs" % (time.time() - t0))
