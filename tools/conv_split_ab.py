"""A/B of the split-K form of the patch-staged 3x3 kernel: 1024 -> 1024 @32x64, forward and reflect adjoint,
timed with HIP events over 50 back-to-back launches (weights rotate over 4 copies so that L2 does not flatter it).
Run once with IR2RGB_CONV3X3P_SPLIT=0 and once with =1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import conv as C
dev = torch.device("cuda:0")
dt = torch.bfloat16
for (cin, h, w) in ((1024, 32, 64), (512, 64, 128)):
    x = torch.randn(1, cin, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    for adj in (False, True):
        d = C.make_desc(x.shape, cin, 3, 1, 1, C.PAD_REFLECT_ADJ if adj else C.PAD_REFLECT, dt)
        wps = []
        for i in range(4):
            wt = torch.randn(cin, cin, 3, 3, device=dev) * 0.02
            wps.append(C.pack_weight(C.make_desc(x.shape, cin, 3, 1, 1, 0, dt), wt, adjoint=True) if adj else C.pack_weight(d, wt))
        y = C.empty_nhwc(1, cin, h, w, dt, dev)
        for i in range(8):
            C.conv2d_fwd(d, x, wps[i % 4], None, want_stats=not adj, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(50):
            C.conv2d_fwd(d, x, wps[i % 4], None, want_stats=not adj, out=y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        fl = 2.0 * h * w * cin * cin * 9
        print("SPLIT=%s %d->%d @%dx%d %s: %.1f us  %.0f TFLOP/s (%.1f %% of 2.5 PF)  ws %d B" % (
            os.environ.get("IR2RGB_CONV3X3P_SPLIT", "1"), cin, cin, h, w, "adjoint" if adj else "forward", us, fl / us / 1e6,
            fl / us / 1e6 / 25.0, C._fwd_workspace(d, x)[1]))
