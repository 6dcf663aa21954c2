"""Training windows only (the bench's workload without its extra measurements): for rocprofv3 --kernel-trace --stats.
    python tools/prof_train.py [windows]   (default 30; the first 12 are the bench's pre-roll)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2, resident_inputs=True)
A, B = V.synthetic_sequence(n + 2, 512, 1024, 1234, dev)
import gc
for i in range(n):
    if i == 14 and os.environ.get("GC_TUNE", "0") == "1":
        gc.collect(); gc.freeze(); gc.set_threshold(50000, 20, 20)
    if i == n - 10:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print("last 10 windows: host issue %.2f ms/window, until the GPU is done %.2f ms/window" % (t_issue * 100, (time.perf_counter() - t0) * 100))
