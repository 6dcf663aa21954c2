"""The low-channel / full-resolution convolutions of the north-star forward, timed alone (HIP events, 20 runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import conv as C
dev = torch.device("cuda:0")
dt = torch.bfloat16
SHAPES = [  # Cin, Hin, Win, Cout, k, stride, pad, pad_mode, transposed, opad, out_f32
    (256, 256, 512, 128, (3, 3), 2, 1, 0, True, 1, False),
    (128, 512, 1024, 24, (1, 7), 1, (0, 3), 1, False, 0, True),
    (64, 512, 1024, 128, (7, 1), 1, (3, 0), 1, False, 0, False),
    (128, 512, 1024, 256, (3, 3), 2, 1, 0, False, 0, False),
    (512, 128, 256, 256, (3, 3), 2, 1, 0, True, 1, False),
    (1024, 64, 128, 512, (3, 3), 2, 1, 0, True, 1, False),
    (256, 256, 512, 512, (3, 3), 2, 1, 0, False, 0, False),
    (512, 128, 256, 1024, (3, 3), 2, 1, 0, False, 0, False),
]
for cin, h, w, cout, k, st, pad, pm, tr, opad, f32 in SHAPES:
    x = torch.randn(1, cin, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    d = C.make_desc(tuple(x.shape), cout, k, st, pad, pm, dt, tr, opad)
    if f32:
        d.out_f32 = 1
    wt = torch.randn((cin, cout) + k if tr else (cout, cin) + k, device=dev) * 0.02
    wp = C.pack_weight(d, wt)
    for _ in range(3):
        y, _ = C.conv2d_fwd(d, x, wp, None, want_stats=not f32)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        C.conv2d_fwd(d, x, wp, None, want_stats=not f32)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("%-46s %8.1f us %8.1f TFLOP/s  %s" % ((cin, h, w, cout, k, st, tr), us, C._flops(d) / us / 1e6, C.kernel_name(d)), flush=True)
