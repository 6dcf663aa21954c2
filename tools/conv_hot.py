"""Runs only the hot 3x3 reflect 1024->1024 @64x128 convolution (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import conv as C
dev = torch.device("cuda:0")
dt = torch.bfloat16
cin = cout = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 128)
x = torch.randn(1, cin, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
adj = len(sys.argv) > 4 and sys.argv[4] == "adj"     # the in-place reflect adjoint (data gradient) of the same layer
d = C.make_desc(x.shape, cout, 3, 1, 1, C.PAD_REFLECT_ADJ if adj else C.PAD_REFLECT, dt)
wp = C.pack_weight(C.make_desc(x.shape, cout, 3, 1, 1, 0, dt), wt, adjoint=True) if adj else C.pack_weight(d, wt)
y = C.empty_nhwc(1, cout, h, w, dt, dev)
b = torch.zeros(cout, device=dev)
for _ in range(20):
    C.conv2d_fwd(d, x, wp, None if adj else b, want_stats=not adj, out=y)
torch.cuda.synchronize()
print("done")
