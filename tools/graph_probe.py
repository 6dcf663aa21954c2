"""Feasibility / payoff probe: capture one steady-state training window (as is) into a HIP graph and time its replay.
State that moves between windows (frame-history positions, Adam's step count) is frozen at the captured values here:
the numbers are timing only."""
import os, sys, time
os.environ.setdefault("IR2RGB_FLOW_STREAM", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2)
tr.flow_net.use_graph = False
A, B = V.synthetic_sequence(40, 512, 1024, 1234, dev)
for i in range(12):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(12, 22):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
print("eager (FlowNet2 inline, one stream): %.2f ms/window" % ((time.perf_counter() - t0) * 100), flush=True)
gA, gB = A[:, 22:25].clone(), B[:, 22:25].clone()
for o in [tr.optimizer_G, tr.optimizer_D] + tr.optimizer_D_T:
    o._copied = [None, None]
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    out = tr.train_window(gA, gB)
torch.cuda.synchronize()
print("captured", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("graph replay: %.2f ms/window" % ((time.perf_counter() - t0) * 50), flush=True)
print({k: float(v) for k, v in list(out.items())[:4]})
