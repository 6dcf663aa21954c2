"""How much would batching the three discriminator forwards of a window (real, fake, raw) save?  Times netD forward +
backward as 3 calls with batch 1 against 1 call with batch 3 (BatchNorm statistics then span the batch: timing probe only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import networks as N
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (cin, h, w, tag) in ((6, 512, 1024, "image D"), (13, 512, 1024, "temporal D")):
    D = N.build_discriminator_module(cin, 64, 3, "batch", 2, True).to(dev).train()
    x1 = [torch.randn(1, cin, h, w, device=dev, requires_grad=True) for _ in range(3)]
    x3 = torch.randn(3, cin, h, w, device=dev, requires_grad=True)
    def run(xs):
        tot = 0
        for x in xs:
            out = D(x)
            tot = tot + sum(t.float().mean() for scale in out for t in scale)
        tot.backward()
    for xs, name in ((x1, "3 x batch 1"), ([x3], "1 x batch 3")):
        for _ in range(3):
            run(xs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(10):
            run(xs)
        e1.record()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("%-11s %-12s GPU %.2f ms  host issue %.2f ms per fwd+bwd" % (tag, name, e0.elapsed_time(e1) / 10, t_issue * 100))
