"""How much of a training window is host issue time?  Times 10 steady-state windows: the host's time to ISSUE them (no
synchronisation) against the time until the GPU has finished them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2)
A, B = V.synthetic_sequence(32, 512, 1024, 1234, dev)
for i in range(14):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(14, 24):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("10 windows: host issue %.1f ms/window, until the GPU is done %.1f ms/window" % (t_issue * 100, t_all * 100))
