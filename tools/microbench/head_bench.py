import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ir2rgb_amd import conv as C
dev = torch.device("cuda:0")
for cin, h, w in ((128, 512, 1024), (64, 512, 1024), (64, 1024, 2048)):
    x = torch.randn(1, cin, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = torch.randn(24, cin, 1, 7, device=dev) * 0.02
    d = C.make_desc(tuple(x.shape), 24, (1, 7), 1, (0, 3), C.PAD_REFLECT, torch.bfloat16, out_f32=True)
    wp = C.pack_weight(d, wt)
    y = C.empty_nhwc(1, 24, h, w, torch.float32, dev)
    for _ in range(3):
        C.conv2d_fwd(d, x, wp, out=y)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        C.conv2d_fwd(d, x, wp, out=y)
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / 50 * 1e3
    mb = (x.numel() * 2 + y.numel() * 4) / 1e6
    print(C.kernel_name(d), (cin, h, w), "%.1f us  %.0f MB -> %.2f TB/s" % (us, mb, mb / us / 1e6 * 1e6 / 1e6))
