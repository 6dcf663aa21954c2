import time, torch, torch.nn.functional as F
dev = torch.device("cuda:0")
def run(name, cin, cout, h, w, k, s, p, dt):
    x = torch.randn(1, cin, h, w, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last).requires_grad_()
    wt = (torch.randn(cout, cin, k, k, device=dev, dtype=dt) * 0.02).contiguous(memory_format=torch.channels_last).requires_grad_()
    t0 = time.time()
    y = F.conv2d(x, wt, None, s, p); y.backward(torch.ones_like(y)); torch.cuda.synchronize()
    first = time.time() - t0
    def step():
        y = F.conv2d(x, wt, None, s, p); y.backward(y)
    for _ in range(3): step()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): step()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 10 * 1e-3
    with torch.no_grad():
        for _ in range(3): F.conv2d(x, wt, None, s, p)
        torch.cuda.synchronize(); a.record()
        for _ in range(10): F.conv2d(x, wt, None, s, p)
        b.record(); torch.cuda.synchronize()
    tf = a.elapsed_time(b) / 10 * 1e-3
    fl = 2.0 * (h // s) * (w // s) * cout * cin * k * k
    print(f"{name} {dt}: first call {first:.1f}s  fwd {tf*1e6:.0f}us ({fl/tf/1e12:.0f} TF)  fwd+bwd {t*1e6:.0f}us ({3*fl/t/1e12:.0f} TF)", flush=True)
for dt in (torch.bfloat16,):
    run("res1024@64x128", 1024, 1024, 64, 128, 3, 1, 1, dt)
    run("loc128@256x512", 128, 128, 256, 512, 3, 1, 1, dt)
    run("down512->1024", 512, 1024, 128, 256, 3, 2, 1, dt)
    run("D64->128", 64, 128, 257, 513, 4, 2, 2, dt)
