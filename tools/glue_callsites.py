"""Which call sites issue the torch glue ops of a training window (contiguous copies, cat, zeros, zero_)?
Wraps the Python entry points and records caller + bytes; no profiler needed."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2, resident_inputs=True)
A, B = V.synthetic_sequence(24, 512, 1024, 1234, dev)
for i in range(14):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0])
def site():
    for f in reversed(traceback.extract_stack(limit=8)[:-2]):
        if "ir2rgb_amd" in f.filename:
            return "%s:%d" % (f.filename.split("ir2rgb_amd/")[-1], f.lineno)
    return "?"
def rec(kind, nbytes):
    a = agg[(kind, site())]; a[0] += 1; a[1] += nbytes
oc = torch.Tensor.contiguous
def contiguous(self, *a, **k):
    r = oc(self, *a, **k)
    if r.data_ptr() != self.data_ptr() or r is not self and r._base is None and not self.is_contiguous(*a, **k):
        rec("contiguous", r.numel() * r.element_size())
    return r
torch.Tensor.contiguous = contiguous
of = torch.Tensor.float
def tfloat(self, *a, **k):
    r = of(self, *a, **k)
    if r is not self: rec("float", r.numel() * 4)
    return r
torch.Tensor.float = tfloat
for name in ("cat", "stack", "zeros", "zeros_like"):
    def mk(name, orig):
        def w(*a, **k):
            r = orig(*a, **k)
            if r.is_cuda: rec(name, r.numel() * r.element_size())
            return r
        return w
    setattr(torch, name, mk(name, getattr(torch, name)))
for name in ("zero_", "new_zeros", "clone"):
    def mk(name, orig):
        def w(self, *a, **k):
            r = orig(self, *a, **k)
            if r.is_cuda: rec(name, r.numel() * r.element_size())
            return r
        return w
    setattr(torch.Tensor, name, mk(name, getattr(torch.Tensor, name)))
tr.train_window(A[:, 15:18], B[:, 15:18])
torch.cuda.synchronize()
for (k, s), (n, b) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:50]:
    print("%-11s %4d %9.2f MB  %s" % (k, n, b / 1e6, s))
