"""Determinism probe of the two-stream generator forward: repeated forwards with IR2RGB_BRANCH_STREAMS on must be
bit-identical to the single-stream result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import networks as N
dev = torch.device("cuda:0")
opt = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)
H, W = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
g1 = N.build_generator_module(9, 3, 6, 64, "composite-local", 3, "batch", 1, **opt).to(dev).train()
g1.compute_dtype = torch.float16
gen = torch.Generator().manual_seed(1)
A, P = torch.rand(1, 9, H, W, generator=gen).to(dev), torch.rand(1, 6, H, W, generator=gen).to(dev)
fi = torch.rand(1, 128, H // 2, W // 2, generator=gen).to(dev).half().contiguous(memory_format=torch.channels_last)
ff = torch.rand(1, 128, H // 2, W // 2, generator=gen).to(dev).half().contiguous(memory_format=torch.channels_last)
def run(mode):
    N.BRANCH_STREAMS = mode
    with torch.no_grad():
        out = g1(A, P, None, fi, ff, None, False)
    torch.cuda.synchronize()
    return [o.clone() for o in out[:6]]
ref = run("0")
ref2 = run("0")
print("single-stream repeat identical:", all(torch.equal(a, b) for a, b in zip(ref, ref2)))
names = ["final", "flow", "weight", "raw", "img_feat", "flow_feat"]
for it in range(8):
    out = run("1")
    diff = {n: int((a != b).sum()) for n, a, b in zip(names, out, ref) if not torch.equal(a, b)}
    print("two-stream run", it, "elements differing:", diff)

# ---- first forward of a FRESH module (weights packed, caches created while the two streams run)
import copy
from ir2rgb_amd import layers as L
for it in range(4):
    L._UNIT.clear()
    torch.manual_seed(0)
    ga = N.build_generator_module(9, 3, 6, 64, "composite-local", 3, "batch", 1, **opt).to(dev).train()
    gb = copy.deepcopy(ga)
    ga.compute_dtype = gb.compute_dtype = torch.float16
    torch.cuda.synchronize()
    N.BRANCH_STREAMS = "1"
    with torch.no_grad():
        oa = ga(A, P, None, fi, ff, None, False)
    torch.cuda.synchronize()
    N.BRANCH_STREAMS = "0"
    with torch.no_grad():
        ob = gb(A, P, None, fi, ff, None, False)
    torch.cuda.synchronize()
    diff = {n: int((a != b).sum()) for n, a, b in zip(names, oa[:6], ob[:6]) if not torch.equal(a, b)}
    print("fresh module, first forward two-stream vs single-stream:", diff)
