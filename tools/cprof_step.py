"""Host-side profile of the training window: cProfile over 3 steady-state windows, top functions by own time and by
cumulative time (per window)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2)
A, B = V.synthetic_sequence(20, 512, 1024, 1234, dev)
for i in range(12):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(12, 15):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
