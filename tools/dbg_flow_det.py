import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
fn = V.FlowNet(torch.bfloat16, use_graph=False).to(dev)
A, B = V.synthetic_sequence(12, 64, 128, 3, dev)
fr = B[0]                                   # [12, 3, H, W]
for n in (1, 2, 3):
    a, b = fr[3:3 + n].contiguous(), fr[0:n].contiguous()
    ref = None
    bad = 0
    for it in range(30):
        fl, cf = fn(a, b)
        torch.cuda.synchronize()
        if ref is None: ref = fl.clone()
        elif not torch.equal(ref, fl): bad += 1; d = (ref - fl).abs().max().item()
    print("N=%d eager: %d of 29 repeats differ%s" % (n, bad, (" (max %.3g)" % d) if bad else ""))
a, b = fr[3:6].contiguous(), fr[0:3].contiguous()
f3, _ = fn(a, b)
f2, _ = fn(a[:2].contiguous(), b[:2].contiguous())
f1, _ = fn(a[:1].contiguous(), b[:1].contiguous())
print("sample 0: N=3 vs N=2 maxdiff %.3g, N=3 vs N=1 %.3g; sample 1: N=3 vs N=2 %.3g" % ((f3[0] - f2[0]).abs().max().item(), (f3[0] - f1[0]).abs().max().item(), (f3[1] - f2[1]).abs().max().item()))
