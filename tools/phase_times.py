"""GPU time per phase of a steady-state training window at 512x1024 (HIP events at the phase boundaries, mean of 10)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2)
A, B = V.synthetic_sequence(40, 512, 1024, 1234, dev)
marks = []
def ev(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def g(*a, **k):
        ev("<" + name); r = f(*a, **k); ev(">" + name); return r
    setattr(obj, attr, g)
for attr in ("reference_flows", "generate", "image_losses", "skipped_frames", "temporal_losses", "optimizer_steps"):
    wrap(tr, attr, attr)
orig_bp = tr.backward_passes
def bp(loss_G, loss_D, loss_D_T, g_inputs=None):
    # same statements as Vid2VidTrainer.backward_passes with marks between the passes
    from ir2rgb_amd import autograd
    self = tr
    ev("<zero"); self.grads_G.zero(); self.grads_D.zero()
    for gdt in self.grads_DT: gdt.zero()
    ev(">zero")
    shared = self.opt["shared_fake_forward"]
    d_nets = [self.netD] + self.netD_T
    if g_inputs is None and shared: g_inputs = self.grads_G.params
    ev("<backward_G")
    batched = shared and self.opt["batched_D"]
    with autograd.backward_flags([self.netD] if shared else [], autograd.SKIP_PARAM_GRADS, 2 if batched else None), \
            autograd.backward_flags(self.netD_T if shared else [], autograd.SKIP_PARAM_GRADS, 1 if batched else None):
        loss_G.backward(retain_graph=shared, inputs=g_inputs)
    ev(">backward_G")
    self.grads_G.all_reduce_async(self.world)
    with autograd.backward_flags(d_nets if shared else [], autograd.SKIP_INPUT_GRAD):
        ev("<backward_D"); loss_D.backward(inputs=self.grads_D.params if shared else None); ev(">backward_D")
        self.grads_D.all_reduce_async(self.world)
        for s, ld in enumerate(loss_D_T):
            ev("<backward_DT"); ld.backward(inputs=self.grads_DT[s].params if shared else None); ev(">backward_DT")
            self.grads_DT[s].all_reduce_async(self.world)
tr.backward_passes = bp
for i in range(14):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
tot = collections.OrderedDict()
n = 10
for i in range(14, 14 + n):
    marks.clear()
    ev("<window")
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
    ev(">window")
    torch.cuda.synchronize()
    open_ = {}
    for name, e in marks:
        if name[0] == "<": open_[name[1:]] = e
        else: tot[name[1:]] = tot.get(name[1:], 0.0) + open_[name[1:]].elapsed_time(e)
for k, v in tot.items():
    print("%-18s %7.2f ms" % (k, v / n))
print("sum of phases      %7.2f ms" % (sum(v for k, v in tot.items() if k != "window") / n))
