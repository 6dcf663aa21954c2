"""Steady state, VARYING inputs: two-stream forward vs single-stream forward of the same module on the same input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import networks as N
dev = torch.device("cuda:0")
opt = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)
model = sys.argv[1]
H, W = int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
local = model == "composite-local"
g = N.build_generator_module(9, 3, 6, 64 if local else 128, model, 3, "batch", 1 if local else 0, **opt).to(dev).train()
g.compute_dtype = torch.float16
gen = torch.Generator().manual_seed(1)
names6 = ["final", "flow", "weight", "raw", "img_feat", "flow_feat"]
bad = 0
for it in range(12):
    A, P = torch.rand(1, 9, H, W, generator=gen).to(dev) * (1 + it), torch.rand(1, 6, H, W, generator=gen).to(dev)
    fi = ff = None
    if local:
        fi = (torch.rand(1, 128, H // 2, W // 2, generator=gen) * (it + 1)).to(dev).half().contiguous(memory_format=torch.channels_last)
        ff = torch.rand(1, 128, H // 2, W // 2, generator=gen).to(dev).half().contiguous(memory_format=torch.channels_last)
    outs = []
    for mode in ("0", "1"):
        N.BRANCH_STREAMS = mode
        with torch.no_grad():
            outs.append([t.clone() for t in g(A, P, None, fi, ff, None, False)[:6]])
        torch.cuda.synchronize()
    d = {n: int((a != b).sum()) for n, a, b in zip(names6, outs[1], outs[0]) if not torch.equal(a, b)}
    bad += bool(d)
    print(model, "iter", it, d)
print("mismatching iterations:", bad)
