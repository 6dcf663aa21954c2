"""Times the MFMA convolution at the hot shapes of the 512x1024 generator (GPU box)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import conv as C

SHAPES = [  # name, Cin, H, W, Cout, k, stride, pad, pad_mode, transposed, opad
    ("res3x3_1024@64x128", 1024, 64, 128, 1024, 3, 1, 1, 1, False, 0),
    ("res3x3_1024@32x64", 1024, 32, 64, 1024, 3, 1, 1, 1, False, 0),
    ("local3x3_128@256x512", 128, 256, 512, 128, 3, 1, 1, 1, False, 0),
    ("local3x3_128@512x1024", 128, 512, 1024, 128, 3, 1, 1, 1, False, 0),
    ("down3x3s2_512->1024@128x256", 512, 128, 256, 1024, 3, 2, 1, 0, False, 0),
    ("down3x3s2_128->256@512x1024", 128, 512, 1024, 256, 3, 2, 1, 0, False, 0),
    ("up3x3T_1024->512@64x128", 1024, 64, 128, 512, 3, 2, 1, 0, True, 1),
    ("D4x4s2_64->128@257x513", 64, 257, 513, 128, 4, 2, 2, 0, False, 0),
    # training sizes (G0 at 256x512: residual blocks at 32x64)
    ("up3x3T_1024->512@32x64", 1024, 32, 64, 512, 3, 2, 1, 0, True, 1),
    ("up3x3T_512->256@64x128", 512, 64, 128, 256, 3, 2, 1, 0, True, 1),
    ("up3x3T_256->128@128x256", 256, 128, 256, 128, 3, 2, 1, 0, True, 1),
    ("down3x3s2_512->1024@64x128", 512, 64, 128, 1024, 3, 2, 1, 0, False, 0),
    # sub-pixel-class launches of a training window: G1's up-sampler and the discriminators' data gradients
    ("up3x3T_128->64@256x512", 128, 256, 512, 64, 3, 2, 1, 0, True, 1),
    ("D4x4s2T_128->64@129x257", 128, 129, 257, 64, 4, 2, 2, 0, True, 1),
    ("D4x4s2T_256->128@65x129", 256, 65, 129, 128, 4, 2, 2, 0, True, 1),
    ("D4x4s2T_128->64@65x129", 128, 65, 129, 64, 4, 2, 2, 0, True, 1),
    ("D4x4s2T_256->128@33x65", 256, 33, 65, 128, 4, 2, 2, 0, True, 1),
    ("D4x4s1_512->256@66x130(dgrad)", 512, 66, 130, 256, 4, 1, 1, 0, False, 0),
    ("D4x4s1_512->256@34x66(dgrad)", 512, 34, 66, 256, 4, 1, 1, 0, False, 0),
    ("D4x4s1_64->512@67x131(dgrad)", 64, 67, 131, 512, 4, 1, 1, 0, False, 0),
]
if os.environ.get("ONLY"):
    SHAPES = [s for s in SHAPES if os.environ["ONLY"] in s[0]]


def main():
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if "--f16" not in sys.argv else torch.float16
    iters = int(os.environ.get("ITERS", "100"))
    for name, cin, h, w, cout, k, s, p, pm, tr, op in SHAPES:
        x = torch.randn(1, cin, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        wt = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device=dev) * 0.02
        b = torch.zeros(cout, device=dev)
        d = C.make_desc(x.shape, cout, k, s, p, pm, dt, tr, op)
        wp = C.pack_weight(d, wt)
        y = C.empty_nhwc(1, cout, d.Hout, d.Wout, dt, dev)
        for _ in range(3):
            C.conv2d_fwd(d, x, wp, b, want_stats=True, out=y)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            C.conv2d_fwd(d, x, wp, b, want_stats=True, out=y)
        e.record(); torch.cuda.synchronize()
        t = a.elapsed_time(e) / iters * 1e-3
        flops = 2.0 * d.Hout * d.Wout * cout * cin * k * k / (s * s if tr else 1)
        print(name, json.dumps(dict(us=round(t * 1e6, 1), TFLOPs=round(flops / t / 1e12, 1), frac_of_2500=round(flops / t / 2.5e15, 3))))


if __name__ == "__main__":
    main()
