import sys, os
sys.path.insert(0, os.getcwd())
import torch, streamgen, h264decode_amd as H
s = streamgen.encode(**streamgen.recipe("C3", frames=4, idr_period=4, seed=5))[0]
for i in range(6):
    dec = H.Decoder(max_streams=64, max_width=1920, max_height=1088, max_frames_per_batch=30)
    dec.decode([s] * 4)
    free, total = torch.cuda.mem_get_info()
    print(i, "free GB %.1f" % (free / 2**30))
    dec.close()
free, total = torch.cuda.mem_get_info()
print("after close: free GB %.1f of %.1f" % (free / 2**30, total / 2**30))
