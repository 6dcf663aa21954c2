"""Per-kernel breakdown of the training window with torch.profiler (kineto / roctracer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2, resident_inputs=True)
A, B = V.synthetic_sequence(24, 512, 1024, 1234, dev)
for i in range(16):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
print("warm", flush=True)
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=("--stack" in sys.argv)) as prof:
    for i in range(16, 19):
        tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
    torch.cuda.synchronize()
tab = prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70)
open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "torchprof_step.txt"), "w").write(tab)
print(tab[-6000:])
tab2 = prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60)
open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "torchprof_step_cpu.txt"), "w").write(tab2)
if "--stack" in sys.argv:
    tab3 = prof.key_averages(group_by_stack_n=6).table(sort_by="self_cpu_time_total", row_limit=60, max_name_column_width=50, max_src_column_width=110)
    open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "torchprof_step_stack.txt"), "w").write(tab3)
