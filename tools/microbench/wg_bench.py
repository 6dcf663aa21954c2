import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ir2rgb_amd import conv as C
dev = torch.device("cuda:0")
dt = torch.bfloat16
for (n, cin, h, w, cout, k, s, p, pm) in ((1, 64, 512, 1024, 64, (7, 1), (1, 1), (3, 0), 1), (1, 64, 512, 1024, 64, (1, 7), (1, 1), (0, 3), 1),
                                          (1, 64, 256, 512, 128, (7, 1), (1, 1), (3, 0), 1), (1, 128, 256, 512, 64, (1, 7), (1, 1), (0, 3), 1),
                                          (3, 64, 512, 513, 64, (4, 1), (2, 1), (2, 0), 0), (3, 64, 256, 257, 64, (4, 1), (2, 1), (2, 0), 0)):
    x = torch.randn(n, cin, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    d = C.make_desc(tuple(x.shape), cout, k, s, p, pm, dt)
    gy = torch.randn(n, cout, d.Hout, d.Wout, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    for _ in range(3): C.conv2d_wgrad(d, x, gy)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): C.conv2d_wgrad(d, x, gy)
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / 30 * 1e3
    fl = 2.0 * n * d.Hout * d.Wout * cin * cout * k[0] * k[1]
    print((n, cin, h, w, cout, k, s), "%.1f us  %.0f TFLOP/s" % (us, fl / us / 1e6))
