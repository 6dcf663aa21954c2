import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ir2rgb_amd import layers as L
dev = torch.device("cuda:0")
for rows, ch in ((16384, 64), (8192, 64), (4096, 64), (4096, 128), (2048, 128)):
    stats = torch.rand(rows, 2, ch, device=dev)
    bn = torch.nn.BatchNorm2d(ch).to(dev)
    for _ in range(3): L.bn_finalize(stats, rows * 128, bn, True, None)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): L.bn_finalize(stats, rows * 128, bn, True, None)
    e.record(); torch.cuda.synchronize()
    print((rows, ch), "%.1f us" % (a.elapsed_time(e) * 20))
