import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
A, B = V.synthetic_sequence(6, 128, 256, 7, dev)
for fused in (True, False):
    tr = V.Vid2VidTrainer(dev, seed=0, first_layer_gen_filters=64, gen_blocks=2, fused_adam=fused)
    opt = tr.optimizer_D
    params = tr.grads_D.params
    names = [n for n, p in tr.netD.named_parameters() if p.requires_grad]
    shadow = [torch.nn.Parameter(p.detach().clone()) for p in params]
    sh_opt = torch.optim.Adam(shadow, lr=2e-4, betas=(0.5, 0.999))
    real_step = opt.step
    def step():
        torch.cuda.synchronize()
        for s, p in zip(shadow, params):
            s.grad = p.grad.detach().clone()
        gsum = sum(p.grad.double().abs().sum().item() for p in params)
        real_step()
        sh_opt.step()
        torch.cuda.synchronize()
        worst = max(((p - s).abs().max().item(), n) for n, p, s in zip(names, params, shadow))
        print("fused" if fused else "torch", "grad abs sum %.6e" % gsum, "worst |p - shadow|", worst)
    opt.step = step
    for w in range(2):
        out = tr.train_window(A[:, w:w + 3], B[:, w:w + 3])
        print(w, {k: round(v.item(), 5) for k, v in out.items()})
