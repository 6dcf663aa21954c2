"""Attribute the GPU time of torch's own (aten) kernels in a training window to the Python line that issued them."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2)
A, B = V.synthetic_sequence(20, 512, 1024, 1234, dev)
for i in range(10):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
NW = 2
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    for i in range(10, 10 + NW):
        tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    p = e.cpu_parent
    if p is not None and p.name.startswith("aten::"):
        continue                      # count only the outermost aten op
    t = e.device_time_total
    if t <= 0:
        continue
    fr = [s for s in e.stack if "ir2rgb_amd" in s or "bench.py" in s][:1]
    where = fr[0].split("ir2rgb_amd/")[-1] if fr else ("<%s> %s" % (p.name if p is not None else "top", str(e.input_shapes)[:90]))
    a = agg[(e.name, where)]
    a[0] += t; a[1] += 1
tot = sum(a[0] for a in agg.values())
print("aten device time per window: %.1f us in %d ops" % (tot / NW, sum(a[1] for a in agg.values()) / NW))
for (n, w), a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print("%8.1f %5.1f  %-22s %s" % (a[0] / NW, a[1] / NW, n, w[:120]))
