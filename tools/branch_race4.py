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
names6 = ["final", "flow", "weight", "raw", "img_feat", "flow_feat"]
ref0 = None
modes = sys.argv[1].split(",")
for it in range(4):
    torch.manual_seed(0)
    g = N.build_generator_module(9, 3, 6, 64, "composite-local", 3, "batch", 1, **opt).to(dev).train()
    g.compute_dtype = torch.float16
    torch.cuda.synchronize()
    N.BRANCH_STREAMS = modes[it % len(modes)]
    with torch.no_grad():
        o = [t.clone() for t in g(Ain, P, None, fi, ff, None, False)[:6]]
    torch.cuda.synchronize()
    if ref0 is None:
        ref0 = o
    out = {}
    for n, a, b in zip(names6, o, ref0):
        if not torch.equal(a, b):
            idx = (a != b).nonzero()
            out[n] = (int(idx.shape[0]), idx[0].tolist(), idx[-1].tolist(), float((a.float() - b.float()).abs().max()))
    print("iter", it, "mode", N.BRANCH_STREAMS, out)
