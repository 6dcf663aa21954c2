import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import vid2vid as V, autograd as A, conv as C
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2)
Aa, B = V.synthetic_sequence(12, 512, 1024, 1234, dev)
for i in range(4):
    tr.train_window(Aa[:, i:i + 3], B[:, i:i + 3])
orig = A._as_half_nhwc
seen = collections.Counter()
def patched(g, dtype):
    if not (g.dtype == dtype and C.is_nhwc(g)):
        seen[(tuple(g.shape), tuple(g.stride()), str(g.dtype))] += 1
    return orig(g, dtype)
A._as_half_nhwc = patched
oc = V._SplitGroupsFn.backward
def bw(ctx, *grads):
    for g in grads:
        if g is not None:
            seen[("split-in", tuple(g.shape), tuple(g.stride()), str(g.dtype))] += 1
    return oc(ctx, *grads)
V._SplitGroupsFn.backward = staticmethod(bw)
tr.train_window(Aa[:, 5:8], B[:, 5:8])
torch.cuda.synchronize()
for k, v in sorted(seen.items(), key=lambda kv: str(kv[0])):
    print(v, k)
