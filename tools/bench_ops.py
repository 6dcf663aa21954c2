"""Micro-benchmark of the three FlowNet2 operators at BASELINE config 3 sizes (GPU box).
Prints one line per op: time, algorithmic GB/s, GFLOP/s.  HIP-event timing on the current stream."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd.ext import correlation_cuda, resample2d_cuda, channelnorm_cuda, warp_diff_norm


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    dev = torch.device("cuda:0")
    res = {}
    f1 = torch.randn(1, 256, 64, 128, device=dev); f2 = torch.randn_like(f1)
    out = torch.empty(1, 441, 64, 128, device=dev); e = torch.empty(0, device=dev)
    t = timeit(lambda: correlation_cuda.forward(f1, f2, e, e, out, 20, 1, 20, 1, 2, 1))
    by = 4 * (2 * f1.numel() + out.numel()); fl = 2 * out.numel() * 256
    res["correlation_fwd"] = dict(us=t * 1e6, GBps=by / t / 1e9, GFLOPs=fl / t / 1e9)
    go = torch.randn_like(out); g1 = torch.empty_like(f1); g2 = torch.empty_like(f1)
    t = timeit(lambda: correlation_cuda.backward(f1, f2, e, e, go, g1, g2, 20, 1, 20, 1, 2, 1), iters=5, warm=1)
    res["correlation_bwd"] = dict(us=t * 1e6, GFLOPs=2 * fl / t / 1e9)
    img = torch.randn(1, 3, 512, 1024, device=dev); flow = torch.randn(1, 2, 512, 1024, device=dev) * 4
    o = torch.empty_like(img)
    t = timeit(lambda: resample2d_cuda.forward(img, flow, o, 1))
    res["resample2d_fwd"] = dict(us=t * 1e6, GBps=4 * (8 * 512 * 1024) / t / 1e9)
    n = torch.empty(1, 1, 512, 1024, device=dev)
    t = timeit(lambda: channelnorm_cuda.forward(img, n, 2))
    res["channelnorm_fwd"] = dict(us=t * 1e6, GBps=4 * (4 * 512 * 1024) / t / 1e9)
    t = timeit(lambda: warp_diff_norm(img, img, flow))
    res["warp_diff_norm(3 outs)"] = dict(us=t * 1e6, GBps=4 * (15 * 512 * 1024) / t / 1e9)
    for k, v in res.items():
        print(k, json.dumps({a: round(b, 1) for a, b in v.items()}))


if __name__ == "__main__":
    main()
