"""Which call sites issue the torch (aten) kernels of a training window?  torch.profiler with stacks; every aten op that
launched device kernels is attributed to the innermost ir2rgb_amd frame of its stack (backward ops: to the autograd
function's backward).  Prints count per window, device time per window, by (op, site)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2)
A, B = V.synthetic_sequence(20, 512, 1024, 1234, dev)
for i in range(12):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    for i in range(12, 12 + N):
        tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    dt = e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
    if dt <= 0:
        continue
    site = "?"
    for fr in (e.stack or []):
        if "ir2rgb_amd/" in fr:
            site = fr.split("ir2rgb_amd/")[-1].strip()
            break
    a = agg[(e.name, site)]
    a[0] += 1
    a[1] += dt
tot_n = sum(v[0] for v in agg.values()) / N
tot_t = sum(v[1] for v in agg.values()) / N / 1e3
print("aten ops with device time: %.0f per window, %.2f ms per window" % (tot_n, tot_t))
for (name, site), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%-28s %6.1f/win %8.3f ms/win  %s" % (name, n / N, t / N / 1e3, site[:110]))
