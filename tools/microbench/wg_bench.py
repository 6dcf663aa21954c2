import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ir2rgb_amd import conv as C
dev = torch.device("cuda:0")
dt = torch.bfloat16
SHAPES_S2 = ((1, 64, 512, 1024, 128, (3, 3), (2, 2), (1, 1), 0, False), (1, 128, 256, 512, 256, (3, 3), (2, 2), (1, 1), 0, False),
             (1, 256, 128, 256, 512, (3, 3), (2, 2), (1, 1), 0, False), (1, 512, 64, 128, 1024, (3, 3), (2, 2), (1, 1), 0, False),
             (1, 1024, 32, 64, 512, (3, 3), (2, 2), (1, 1), 0, True), (1, 512, 64, 128, 256, (3, 3), (2, 2), (1, 1), 0, True),
             (1, 256, 128, 256, 128, (3, 3), (2, 2), (1, 1), 0, True), (1, 128, 256, 512, 64, (3, 3), (2, 2), (1, 1), 0, True))
for (n, cin, h, w, cout, k, s, p, pm, tr) in SHAPES_S2:
    x = torch.randn(n, cin, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    d = C.make_desc(tuple(x.shape), cout, k, s, p, pm, dt, tr, 1 if tr else 0)
    gy = torch.randn(n, cout, d.Hout, d.Wout, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    for _ in range(3): C.conv2d_wgrad(d, x, gy)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): C.conv2d_wgrad(d, x, gy)
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / 30 * 1e3
    fl = 2.0 * n * (h * w if tr else d.Hout * d.Wout) * cin * cout * 9
    print('s2', (cin, h, w, cout, 'T' if tr else ''), '%.1f us  %.0f TFLOP/s' % (us, fl / us / 1e6))
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
