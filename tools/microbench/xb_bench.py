import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ir2rgb_amd import autograd as A
dev = torch.device("cuda:0")
for (cin, H, W, kw, sx, px, pm) in ((6, 512, 1024, 4, 2, 2, 0), (15, 512, 1024, 4, 2, 2, 0), (6, 256, 512, 4, 2, 2, 0), (9, 512, 1024, 7, 1, 3, 1)):
    wout = (W + 2 * px - kw) // sx + 1
    d = torch.randn(1, 64, H, wout, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    for _ in range(3): A.xexpand_bwd(d, cin, W, kw, sx, px, pm)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): A.xexpand_bwd(d, cin, W, kw, sx, px, pm)
    e.record(); torch.cuda.synchronize()
    print((cin, H, W, kw, sx), "%.1f us" % (a.elapsed_time(e) * 20))
