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
names6 = ["final", "flow", "weight", "raw", "img_feat", "flow_feat"]
variant = sys.argv[1]
orig_pack = C.pack_weight
def pack_sync(desc, weight, adjoint=False):
    if "pre" in variant: torch.cuda.synchronize()
    out = orig_pack(desc, weight, adjoint)
    if "post" in variant: torch.cuda.synchronize()
    return out
C.pack_weight = pack_sync
orig_fin = L.bn_finalize
def fin_sync(*a, **k):
    out = orig_fin(*a, **k)
    if "bn" in variant: torch.cuda.synchronize()
    return out
L.bn_finalize = fin_sync
ref0 = None
for it in range(4):
    torch.manual_seed(0)
    g = N.build_generator_module(9, 3, 6, 64, "composite-local", 3, "batch", 1, **opt).to(dev).train()
    g.compute_dtype = torch.float16
    torch.cuda.synchronize()
    N.BRANCH_STREAMS = "0" if it == 0 else "1"
    with torch.no_grad():
        o = [t.clone() for t in g(Ain, P, None, fi, ff, None, False)[:6]]
    torch.cuda.synchronize()
    if ref0 is None:
        ref0 = o
    print(variant, "iter", it, {n: int((a != b).sum()) for n, a, b in zip(names6, o, ref0) if not torch.equal(a, b)})
