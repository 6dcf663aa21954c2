import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)
t0 = time.time()
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2)
print("build", round(time.time() - t0, 1), "s", flush=True)
A, B = V.synthetic_sequence(14, H, W, 1234, dev)
for i in range(12):
    t0 = time.time()
    out = tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
    t_issue = time.time() - t0
    torch.cuda.synchronize()
    print(i, round(time.time() - t0, 4), "s (issued in", round(t_issue, 4), "s)", {k: round(v.item(), 4) for k, v in out.items()}, flush=True)
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
