import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import networks as N, autograd as A, layers as L, conv as C
dev = torch.device("cuda:0")
opt = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)
H, W = 1024, 2048
gen = torch.Generator().manual_seed(1)
Ain, P = torch.rand(1, 9, H, W, generator=gen).to(dev), torch.rand(1, 6, H, W, generator=gen).to(dev)
fi = torch.rand(1, 128, H // 2, W // 2, generator=gen).to(dev).half().contiguous(memory_format=torch.channels_last)
ff = torch.rand(1, 128, H // 2, W // 2, generator=gen).to(dev).half().contiguous(memory_format=torch.channels_last)
orig = A.conv_stage
rec = []
def spy(x, conv, bn, *a, **k):
    z = orig(x, conv, bn, *a, **k)
    rec.append((conv, z))
    return z
A.conv_stage = spy
orig_add = A.add
def spy_add(a, b):
    z = orig_add(a, b)
    rec.append(("add%d" % sum(1 for c, _ in rec if isinstance(c, str)), z))
    return z
A.add = spy_add
runs = []
for it in range(3):
    torch.manual_seed(0)
    g = N.build_generator_module(9, 3, 6, 64, "composite-local", 3, "batch", 1, **opt).to(dev).train()
    g.compute_dtype = torch.float16
    names = {m: n for n, m in g.named_modules()}
    torch.cuda.synchronize()
    N.BRANCH_STREAMS = "0" if it == 0 else "1"
    rec.clear()
    with torch.no_grad():
        g(Ain, P, None, fi, ff, None, False)
    torch.cuda.synchronize()
    runs.append({(c if isinstance(c, str) else names[c]): z for c, z in rec})
for it in (1, 2):
    print("--- fresh module", it, "two-stream vs iteration 0 single-stream")
    for k in runs[0]:
        a, b = runs[it].get(k), runs[0][k]
        if a is None:
            print(k, "missing"); continue
        nd = int((a != b).sum())
        if nd:
            idx = (a != b).nonzero()
            print("  %-40s differs in %9d of %9d  max %.4f  first %s last %s" % (k, nd, a.numel(), float((a.float() - b.float()).abs().max()), idx[0].tolist(), idx[-1].tolist()))
