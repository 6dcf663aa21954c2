"""Main-stream GPU time per phase of a steady-state training window at 512x1024 (HIP events recorded on the main stream at
the phase boundaries, mean of 10 windows; a phase includes the main stream's waits for the side streams it joins)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2, resident_inputs=True)
A, B = V.synthetic_sequence(40, 512, 1024, 1234, dev)
marks = []
def ev(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
def wrap(obj, attr, before=None, after=None):
    f = getattr(obj, attr)
    def g(*a, **k):
        if before: ev(before)
        r = f(*a, **k)
        if after: ev(after)
        return r
    setattr(obj, attr, g)
wrap(tr, "generate", "generate", "losses (image D on main)")
wrap(tr, "get_losses", None, "zero + backward_G (through the Ds, G1, G0)")
wrap(tr.grads_G, "all_reduce_async", "backward_D (three discriminators)", None)
wrap(tr, "optimizer_steps", "optimizer steps + repack", "end")
for i in range(16):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
tot = collections.OrderedDict()
n = 10
for i in range(16, 16 + n):
    marks.clear()
    ev("reference flows issue / bookkeeping")
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
    torch.cuda.synchronize()
    for (name, e0), (_, e1) in zip(marks[:-1], marks[1:]):
        tot[name] = tot.get(name, 0.0) + e0.elapsed_time(e1)
for k, v in tot.items():
    print("%-48s %7.2f ms" % (k, v / n))
print("%-48s %7.2f ms" % ("sum", sum(tot.values()) / n))
