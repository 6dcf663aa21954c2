"""Generator forward only (hand-written kernels only, no torch convolutions): for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import networks as N
dev = torch.device("cuda:0")
opt = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)
torch.manual_seed(0)
g = N.build_generator_module(9, 3, 6, 128, "composite", 3, "batch", 0, **opt).to(dev).train()
x, p = torch.tanh(torch.randn(1, 9, 512, 1024, device=dev)), torch.tanh(torch.randn(1, 6, 512, 1024, device=dev))
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        out = g(x, p, None, None, None, None, False)
torch.cuda.synchronize()
import time
with torch.no_grad():
    t0 = time.time()
    for _ in range(10):
        out = g(x, p, None, None, None, None, False)
    torch.cuda.synchronize()
print("ok", float(out[0].abs().mean()), "eager forward ms: %.3f" % ((time.time() - t0) * 100))
if "--graph" in sys.argv:
    from ir2rgb_amd.graphs import GraphedForward
    gf = GraphedForward(lambda a, b: g(a, b, None, None, None, None, False)[:4], x, p)
    for _ in range(3):
        gf(x, p)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        gf(x, p)
    torch.cuda.synchronize()
    ms = (time.time() - t0) * 50
    print("graphed forward ms: %.3f  (%.1f TFLOP/s of 6.632 TFLOP = %.1f %% of 2.5 PF)" % (ms, 6.632e3 / ms, 6.632e3 / ms / 25))
