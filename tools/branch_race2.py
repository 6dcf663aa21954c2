"""Bisect the first-forward difference of the two-stream local generator: per-stage outputs, two-stream vs single-stream."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import networks as N, autograd as A, layers as L
dev = torch.device("cuda:0")
opt = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)
H, W = 1024, 2048
gen = torch.Generator().manual_seed(1)
Ain, P = torch.rand(1, 9, H, W, generator=gen).to(dev), torch.rand(1, 6, H, W, generator=gen).to(dev)
fi = torch.rand(1, 128, H // 2, W // 2, generator=gen).to(dev).half().contiguous(memory_format=torch.channels_last)
ff = torch.rand(1, 128, H // 2, W // 2, generator=gen).to(dev).half().contiguous(memory_format=torch.channels_last)
orig = A.conv_stage
rec = None
KEEP = os.environ.get("KEEP", "1") == "1"
def spy(x, conv, bn, *a, **k):
    z = orig(x, conv, bn, *a, **k)
    rec.append((conv, z if KEEP else None))
    return z
A.conv_stage = spy
torch.manual_seed(0)
ga = N.build_generator_module(9, 3, 6, 64, "composite-local", 3, "batch", 1, **opt).to(dev).train()
gb = copy.deepcopy(ga)
ga.compute_dtype = gb.compute_dtype = torch.float16
names = {m: n for n, m in ga.named_modules()}
names.update({m: n for n, m in gb.named_modules()})
torch.cuda.synchronize()
outs = []
final = []
for g, mode in ((ga, "1"), (gb, "0")):
    N.BRANCH_STREAMS = mode
    rec = []
    with torch.no_grad():
        g(Ain, P, None, fi, ff, None, False)
    torch.cuda.synchronize()
    outs.append({names[c]: z for c, z in rec})
    final.append([o.clone() for o in g(Ain, P, None, fi, ff, None, False)[:6]] if False else None)
if not KEEP:
    print("no retention: compare final outputs instead")
    import sys as _s
for k in (outs[1] if KEEP else []):
    a, b = outs[0][k], outs[1][k]
    nd = int((a != b).sum())
    if nd:
        d = (a.float() - b.float()).abs()
        idx = (a != b).nonzero()
        print(k, "differs in", nd, "of", a.numel(), "max abs", float(d.max()), "first", idx[0].tolist(), "last", idx[-1].tolist())
print("done")
