import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2, resident_inputs=True)
A, B = V.synthetic_sequence(8, 512, 1024, 1234, dev)
for i in range(4):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
nG = sum(p.numel() for p in tr.optimizer_G.params)
nD = sum(p.numel() for p in tr.optimizer_D.params)
us = timeit(tr.optimizer_G.step)
print("Adam G: %d tensors, %.1f M params, %.1f us, %.2f TB/s (28 B/param)" % (len(tr.optimizer_G.params), nG / 1e6, us, nG * 28 / us / 1e6))
us = timeit(tr.optimizer_D.step)
print("Adam D: %d tensors, %.1f M params, %.1f us, %.2f TB/s" % (len(tr.optimizer_D.params), nD / 1e6, us, nD * 28 / us / 1e6))
for s, o in enumerate(tr.optimizer_D_T):
    n = sum(p.numel() for p in o.params)
    if any(p.grad is None for p in o.params): continue
    us = timeit(o.step)
    print("Adam D_T%d: %.1f M params, %.1f us, %.2f TB/s" % (s, n / 1e6, us, n * 28 / us / 1e6))
rp = tr.repacker
tot_w = tot_p = 0
for b in rp.batch:
    for w, p in b.keep:
        tot_w += w.numel() * 4; tot_p += p.numel() * p.element_size()
us = timeit(rp.run)
print("repack: %d batches, reads %.1f MB fp32, writes %.1f MB halves, %.1f us, %.2f TB/s" % (len(rp.batch), tot_w / 1e6, tot_p / 1e6, us, (tot_w + tot_p) / us / 1e6))
for b in rp.batch:
    us = timeit(b.run)
    print("  batch: %d jobs %d blocks %.1f us" % (len(b.keep), b.nblocks, us))
