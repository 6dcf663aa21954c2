"""Timing of the MFMA cost volume (ir2rgb_correlation_nhwc_half) at FlowNetC's shape for 512x1024 frames: [N,256,64,128]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import _lib, conv as CV
dev = torch.device("cuda:0")
lib = _lib.lib()
for N in (1, 2):
    for mode in (0, 1):
        a = torch.randn(N, 256, 64, 128, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        b = torch.randn(N, 256, 64, 128, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        out = torch.empty((N, 441, 64, 128), dtype=torch.float32, device=dev) if mode == 0 else \
            torch.zeros((N, 512, 64, 128), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
        args = (CV._p(a), 256, 0, CV._p(b), 256, 0, CV._p(out), mode, 512 if mode else 0, 32 if mode else 0, 0.1, N, 256, 64, 128, 1)
        for _ in range(5):
            lib.ir2rgb_correlation_nhwc_half(*args, _lib.current_stream(a))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            lib.ir2rgb_correlation_nhwc_half(*args, _lib.current_stream(a))
        e1.record()
        torch.cuda.synchronize()
        print("corr_mfma N=%d out_mode=%d: %.1f us" % (N, mode, e0.elapsed_time(e1) / 50 * 1e3))
