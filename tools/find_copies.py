"""Which Python lines of the north-star forward issue device copies (hipMemcpy / aten::copy_ / aten::fill_)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ir2rgb_amd import networks as N
dev = torch.device("cuda:0")
opt = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)
torch.manual_seed(0)
g = N.build_generator_module(9, 3, 6, 128, "composite", 3, "batch", 0, **opt).to(dev).train()
x, p = torch.tanh(torch.randn(1, 9, 512, 1024, device=dev)), torch.tanh(torch.randn(1, 6, 512, 1024, device=dev))
with torch.no_grad():
    for _ in range(3):
        g(x, p, None, None, None, None, False)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
        g(x, p, None, None, None, None, False)
        torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy"):
        st = [s for s in e.stack if "ir2rgb_amd" in s][:2]
        cnt[(e.name, " <- ".join(st))] += 1
for (k, st), n in cnt.most_common(30):
    print(n, k, st)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=50))
