"""Timing ablations of the patch-staged 3x3 kernel (library built with -DIR2RGB_ABLATION; IR2RGB_CONV3X3P_DBG: 1 = no staging after the prologue, 2 = no fragment
reads, 3 = neither; results are garbage, only the time means something).  Shapes: 1024@32x64 (split form) and 1024@64x128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import conv as C
dev = torch.device("cuda:0")
dt = torch.bfloat16
for (cin, h, w) in ((1024, 32, 64), (1024, 64, 128)):
    x = torch.randn(1, cin, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    d = C.make_desc(x.shape, cin, 3, 1, 1, C.PAD_REFLECT, dt)
    wps = [C.pack_weight(d, torch.randn(cin, cin, 3, 3, device=dev) * 0.02) for _ in range(4)]
    y = C.empty_nhwc(1, cin, h, w, dt, dev)
    for i in range(8):
        C.conv2d_fwd(d, x, wps[i % 4], None, want_stats=True, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50):
        C.conv2d_fwd(d, x, wps[i % 4], None, want_stats=True, out=y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("DBG=%s SPLIT=%s %d @%dx%d forward: %.1f us" % (os.environ.get("IR2RGB_CONV3X3P_DBG", "0"), os.environ.get("IR2RGB_CONV3X3P_SPLIT", "1"), cin, h, w, us))
