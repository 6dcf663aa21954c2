"""Runs only the G0-bottleneck weight gradient (1024x1024 3x3 @32x64) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import conv as C
dev = torch.device("cuda:0")
dt = torch.bfloat16
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 64)
x = torch.randn(1, 1024, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
gy = torch.randn(1, 1024, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
d = C.make_desc(x.shape, 1024, 3, 1, 1, 1, dt)
for _ in range(12):
    C.conv2d_wgrad(d, x, gy)
torch.cuda.synchronize()
print("done")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import ctypes
from ir2rgb_amd import _lib
ws = torch.empty(_lib.lib().ir2rgb_conv2d_wgrad_workspace_elems(ctypes.byref(d)), dtype=torch.float32, device=dev)
dw = torch.empty(1024, 1024, 3, 3, device=dev)
lib = _lib.lib()
def run():
    lib.ir2rgb_conv2d_wgrad(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(gy.data_ptr()), ctypes.c_void_p(dw.data_ptr()), ctypes.c_void_p(ws.data_ptr()), _lib.current_stream(x))
for _ in range(3): run()
torch.cuda.synchronize(); a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 20 * 1e-3
fl = 2.0 * h * w * 1024 * 1024 * 9
print(f"wgrad 1024x1024x3x3 @{h}x{w}: {t*1e6:.1f} us  {fl/t/1e12:.0f} TFLOP/s (incl. memset+finish)")
