import os, sys
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.randn(1 << 22, device=dev) * 3
for op in (dist.ReduceOp.AVG, dist.ReduceOp.SUM):
    y = x.clone()
    h = dist.all_reduce(y, op=op, async_op=True); h.wait(); torch.cuda.synchronize()
    print("all_reduce", op, "identity:", torch.equal(x, y), (x - y).abs().max().item())
from ir2rgb_amd import conv as C
gen = torch.Generator().manual_seed(0)
for (cin, h, w, cout, k, stride, pad, pm, tr) in [(256, 16, 32, 256, 3, 1, 1, C.PAD_REFLECT, False), (64, 64, 128, 128, 3, 2, 1, C.PAD_ZERO, False),
                                                  (512, 8, 16, 256, 3, 2, 1, C.PAD_ZERO, True), (128, 32, 64, 128, 3, 1, 1, C.PAD_REFLECT, False)]:
    xx = torch.randn(1, cin, h, w, generator=gen).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    desc = C.make_desc(tuple(xx.shape), cout, k, stride, pad, pm, torch.bfloat16, tr, 1 if tr else 0)
    gy = torch.randn(1, cout, desc.Hout, desc.Wout, generator=gen).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    a = C.conv2d_wgrad(desc, xx, gy)
    flat = torch.zeros(a.numel() + 64, device=dev)
    for off in (0, 4, 12):
        b = C.conv2d_wgrad(desc, xx, gy, out=flat[off:off + a.numel()].view(a.shape))
        print("wgrad", (cin, h, w, cout, k, stride, tr), "offset", off, "equal:", torch.equal(a, b), (a - b).abs().max().item())
dist.destroy_process_group()
