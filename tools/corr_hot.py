"""Runs only the correlation forward at the BASELINE config-3 size (for rocprofv3 --pmc / --kernel-trace passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd.ext import correlation_cuda
dev = torch.device("cuda:0")
f1 = torch.randn(1, 256, 64, 128, device=dev); f2 = torch.randn_like(f1)
out = torch.empty(1, 441, 64, 128, device=dev); e = torch.empty(0, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    correlation_cuda.forward(f1, f2, e, e, out, 20, 1, 20, 1, 2, 1)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
