"""Per-shape table of one steady-state training window at 512x1024: every convolution launch bracketed with HIP events
(conv.PROFILE), grouped by (Cin, Hin, Win, Cout, kh, kw, stride, reflect, transposed) and kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import conv as C
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
os.environ["IR2RGB_FLOW_STREAM"] = "0"      # FlowNet2 inline: the generator forward is then bracketed too
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2)
A, B = V.synthetic_sequence(32, 512, 1024, 1234, dev)
for i in range(14):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
C.PROFILE = {}
tr.train_window(A[:, 14:17], B[:, 14:17])
torch.cuda.synchronize()
rows = []
for key, rec in C.PROFILE["shapes"].items():
    t = sum(a.elapsed_time(b) for a, b in rec["events"])
    rows.append((t, len(rec["events"]), rec["flops"], rec["kernel"], key))
C.PROFILE = None
tot = sum(r[0] for r in rows)
print("bracketed total %.2f ms" % tot)
for t, n, fl, k, key in sorted(rows, key=lambda r: -r[0]):
    print("%-66s %-28s n=%3d avg %7.1f us tot %6.3f ms  %6.1f TF/s" % (str(tuple(v if isinstance(v, str) else int(v) for v in key)), k[:28], n, t / n * 1e3, t, fl * n / t / 1e9))
