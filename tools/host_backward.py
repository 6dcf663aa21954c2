"""Host time of the backward functions (they run in the autograd engine's thread, which cProfile does not see): wall
time spent inside each custom Function's backward per window, and inside a few helpers."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import vid2vid as V, autograd as A, losses as LS, conv as C, layers as L
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2, resident_inputs=True)
Aa, B = V.synthetic_sequence(30, 512, 1024, 1234, dev)
for i in range(14):
    tr.train_window(Aa[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap_static(cls, name, label):
    f = getattr(cls, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        e = acc[label]; e[0] += time.perf_counter() - t0; e[1] += 1
        return r
    setattr(cls, name, staticmethod(g))
for cls in (A.ConvStageFn, A.HeadFn, A.WarpBlendFn, A.AddFn, A.ToHalfFn, A.PadChannelsFn, A.AvgPool3s2Fn, LS._FusedLossFn, V._SplitGroupsFn):
    wrap_static(cls, "backward", cls.__name__ + ".backward")
    wrap_static(cls, "forward", cls.__name__ + ".forward")
def wrap_fn(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        e = acc[mod.__name__.split(".")[-1] + "." + name]; e[0] += time.perf_counter() - t0; e[1] += 1
        return r
    setattr(mod, name, g)
for name in ("bn_bwd", "conv_dgrad", "conv_wgrad", "thin_grad_expand"):
    wrap_fn(A, name)
for name in ("conv2d_fwd", "conv2d_wgrad", "make_desc"):
    wrap_fn(C, name)
for name in ("packed_weight", "bn_finalize_apply", "bn_finalize", "bn_apply"):
    wrap_fn(L, name)
N = 5
t0 = time.perf_counter()
for i in range(14, 14 + N):
    tr.train_window(Aa[:, i:i + 3], B[:, i:i + 3])
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print("host issue %.1f ms/window (with the timers)" % (t_issue / N * 1e3))
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-32s %6.1f calls/win %7.2f ms/win %6.1f us/call" % (k, n / N, t / N * 1e3, t / n * 1e6))
