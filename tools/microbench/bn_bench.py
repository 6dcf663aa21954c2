import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ir2rgb_amd import autograd as A
dev = torch.device("cuda:0")
for C, H, W in ((1024, 32, 64), (512, 33, 65), (256, 33, 65)):
    y = torch.randn(1, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gz = torch.randn(1, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    v = [torch.rand(C, device=dev) + 0.5 for _ in range(4)]
    out = torch.empty_like(y)
    for _ in range(5):
        A.bn_bwd(gz, y, v[0], v[1], v[2], v[3], 1, out=out)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        A.bn_bwd(gz, y, v[0], v[1], v[2], v[3], 1, out=out)
    e.record(); torch.cuda.synchronize()
    print((C, H, W), "%.1f us per bn_bwd call" % (a.elapsed_time(e) * 10))
