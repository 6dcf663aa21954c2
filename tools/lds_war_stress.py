"""Convolutions beside an LDS-heavy kernel on a second stream: every launch must reproduce the solo result bit for bit.

The trigger of the round-2 "two-stream first-forward hazard" in isolation: pack_tile_kernel's
bank-conflicted LDS traffic on the same CU slows a convolution's fragment reads; where a wave could pass its K-step barrier
with such a read still in flight, another wave's LDS-DMA restaged the buffer under it.

    python tools/lds_war_stress.py [path/to/libir2rgb_hip.so] [launches per case]

With the library saved from before the fix (gpurun_tmp/libir2rgb_hip_before_war_fix.so) the 2-stage conv_igemm case fails.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ir2rgb_amd import _lib  # noqa: E402

args = sys.argv[1:]
if args and args[0].endswith(".so"):
    _lib.LIB_PATH = os.path.abspath(args.pop(0))
launches = int(args[0]) if args else 80
from ir2rgb_amd import conv as C  # noqa: E402

dev = torch.device("cuda:0")
print("library:", _lib.LIB_PATH, flush=True)

CASES = {
    # name: (dtype, Cin, H, W, Cout, k, stride, pad, pad_mode)
    "igemm_tp128_128ch_3x3_512x1024": (torch.float16, 128, 512, 1024, 128, 3, 1, 1, C.PAD_REFLECT),
    "igemm_tp256_512ch_3x3s2_128x256": (torch.bfloat16, 512, 128, 256, 1024, 3, 2, 1, C.PAD_ZERO),
    "igemm_4x4s2_64ch_257x513": (torch.bfloat16, 64, 257, 513, 128, 4, 2, 2, C.PAD_ZERO),
    "patch_large_1024ch_64x128": (torch.bfloat16, 1024, 64, 128, 1024, 3, 1, 1, C.PAD_REFLECT),
    "patch_split_1024ch_32x64": (torch.bfloat16, 1024, 32, 64, 1024, 3, 1, 1, C.PAD_REFLECT),
    "patch_adj_split_1024ch_32x64": (torch.bfloat16, 1024, 32, 64, 1024, 3, 1, 1, C.PAD_REFLECT_ADJ),
}


def stress(name, launches, hammer=True):
    dtype, cin, h, w, cout, k, stride, pad, pad_mode = CASES[name]
    gen = torch.Generator().manual_seed(len(name))
    x = torch.randn(1, cin, h, w, generator=gen).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, k, k, generator=gen) * 0.05).to(dev)
    desc = C.make_desc(tuple(x.shape), cout, k, stride, pad, pad_mode, dtype)
    if pad_mode == C.PAD_REFLECT_ADJ:
        wp = C.pack_weight(C.make_desc(tuple(x.shape), cout, k, stride, pad, C.PAD_ZERO, dtype), wt, adjoint=True)
    else:
        wp = C.pack_weight(desc, wt)
    ref, _ = C.conv2d_fwd(desc, x, wp)
    ref2, _ = C.conv2d_fwd(desc, x, wp)
    torch.cuda.synchronize()
    assert torch.equal(ref, ref2), "solo launches differ"
    # the hammer: tiled packing of a 1024 x 1024 x 3 x 3 weight (1024 workgroups, 19 KB LDS each, 2-byte strided reads)
    hw = (torch.randn(1024, 1024, 3, 3, generator=gen) * 0.05).to(dev)
    hdesc = C.make_desc((1, 1024, 32, 64), 1024, 3, 1, 1, C.PAD_ZERO, torch.bfloat16)
    side = torch.cuda.Stream(dev)
    outs = [torch.empty_like(ref) for _ in range(4)]
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    worst = torch.zeros((), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for i in range(launches):
        if hammer:
            with torch.cuda.stream(side):
                for _ in range(3):
                    C.pack_weight(hdesc, hw)
        y, _ = C.conv2d_fwd(desc, x, wp, out=outs[i % 4])
        n = (y != ref).sum()
        bad += (n > 0)
        worst = torch.maximum(worst, n)
    torch.cuda.synchronize()
    print("%-36s %-22s launches %3d  wrong %3d  worst launch: %d elements differ" % (
        name, C.kernel_name(desc), launches, int(bad), int(worst)), flush=True)
    return int(bad)


total = 0
for name in CASES:
    total += stress(name, launches)
print("wrong launches in all:", total)
sys.exit(1 if total else 0)
