"""Per-shape convolution times of the north-star generator forward (HIP events around every launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import networks as N, conv as C
dev = torch.device("cuda:0")
opt = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)
torch.manual_seed(0)
g = N.build_generator_module(9, 3, 6, 128, "composite", 3, "batch", 0, **opt).to(dev).train()
x, p = torch.tanh(torch.randn(1, 9, 512, 1024, device=dev)), torch.tanh(torch.randn(1, 6, 512, 1024, device=dev))
NIT = 5
with torch.no_grad():
    for _ in range(3):
        g(x, p, None, None, None, None, False)
    torch.cuda.synchronize()
    C.PROFILE = {}
    for _ in range(NIT):
        g(x, p, None, None, None, None, False)
    torch.cuda.synchronize()
rows = []
for key, rec in C.PROFILE["shapes"].items():
    ts = [a.elapsed_time(b) * 1e3 for a, b in rec["events"]]
    rows.append((sum(ts) / NIT, len(ts) / NIT, sum(ts) / len(ts), rec["flops"] / (sum(ts) / len(ts)) / 1e6, key, rec["kernel"]))
C.PROFILE = None
print("total conv us/forward: %.1f" % sum(r[0] for r in rows))
print("  us/fwd  calls   us/call  TFLOP/s  (Cin,Hin,Win,Cout,kh,kw,stride,pad_mode,transposed) kernel")
for r in sorted(rows, reverse=True):
    print("%8.1f %6.1f %9.1f %8.1f  %s %s" % r)
