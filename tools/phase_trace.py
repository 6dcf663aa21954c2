"""Kernel totals per phase of the training window.  Run under rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/phase_trace.py run
then   python3 tools/phase_trace.py report out/.../*_kernel_trace.csv
Phases are delimited in the trace by marker kernels (torch special functions nobody else uses)."""
import os, sys, collections, csv
MARK = collections.OrderedDict([("reference_flows", "bessel_j0"), ("generate", "bessel_j1"), ("image_losses", "bessel_y0"),
                                ("temporal_losses", "bessel_y1"), ("backward_G", "modified_bessel_i0"), ("backward_D", "modified_bessel_i1"),
                                ("backward_DT", "modified_bessel_k0"), ("optimizer_steps", "modified_bessel_k1"), ("end", "erfcx")])
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from ir2rgb_amd import vid2vid as V, autograd
    dev = torch.device("cuda:0")
    tr = V.Vid2VidTrainer(dev, n_scales_spatial=2)
    A, B = V.synthetic_sequence(40, 512, 1024, 1234, dev)
    m = torch.ones(64, device=dev)
    on = [False]
    def mark(name):
        if on[0]:
            getattr(torch.special, MARK[name])(m)
    def wrap(attr):
        f = getattr(tr, attr)
        def g(*a, **k):
            mark(attr); return f(*a, **k)
        setattr(tr, attr, g)
    for attr in ("reference_flows", "generate", "image_losses", "temporal_losses", "optimizer_steps"):
        wrap(attr)
    def bp(loss_G, loss_D, loss_D_T, g_inputs=None):
        self = tr
        self.grads_G.zero(); self.grads_D.zero()
        for gdt in self.grads_DT: gdt.zero()
        shared = self.opt["shared_fake_forward"]
        d_nets = [self.netD] + self.netD_T
        if g_inputs is None and shared: g_inputs = self.grads_G.params
        mark("backward_G")
        batched = shared and self.opt["batched_D"]
        with autograd.backward_flags([self.netD] if shared else [], autograd.SKIP_PARAM_GRADS, 2 if batched else None), \
                autograd.backward_flags(self.netD_T if shared else [], autograd.SKIP_PARAM_GRADS, 1 if batched else None):
            loss_G.backward(retain_graph=shared, inputs=g_inputs)
        self.grads_G.all_reduce_async(self.world)
        with autograd.backward_flags(d_nets if shared else [], autograd.SKIP_INPUT_GRAD):
            mark("backward_D"); loss_D.backward(inputs=self.grads_D.params if shared else None)
            self.grads_D.all_reduce_async(self.world)
            mark("backward_DT")
            for s, ld in enumerate(loss_D_T):
                ld.backward(inputs=self.grads_DT[s].params if shared else None)
                self.grads_DT[s].all_reduce_async(self.world)
    tr.backward_passes = bp
    for i in range(14):
        tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
    torch.cuda.synchronize()
    on[0] = True
    for i in range(14, 24):
        tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
        mark("end")
    torch.cuda.synchronize()
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    inv = {v: k for k, v in MARK.items()}
    phase, n_end = None, 0
    tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
    span = collections.defaultdict(int)
    t_phase = None
    for s, e, k in ev:
        hit = [p for f, p in inv.items() if f + "_" in k or f + "<" in k or ("::" + f) in k]
        if hit:
            if phase is not None:
                span[phase] += s - t_phase
            phase = hit[0] if hit[0] != "end" else None
            t_phase = e
            n_end += hit[0] == "end"
            continue
        if phase is not None:
            tot[phase][k][0] += e - s
            tot[phase][k][1] += 1
    n = max(n_end, 1)
    for p in MARK:
        if p not in tot: continue
        ks = tot[p]
        print("== %s: span %.2f ms/window, kernel sum %.2f ms, %d launches" % (p, span[p] / n / 1e6, sum(v[0] for v in ks.values()) / n / 1e6, sum(v[1] for v in ks.values()) // n))
        for k, (t, c) in sorted(ks.items(), key=lambda kv: -kv[1][0])[:14]:
            print("   %-96s %5.1f/win %7.3f ms %7.1f us" % (k[:96], c / n, t / n / 1e6, t / c / 1e3))
