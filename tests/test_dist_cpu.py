"""World-size-2 gloo test (CPU) of the data-parallel exchange step: parameter gradients as views of
one flat buffer, chunked all-reduce, averaged gradients == single-process gradient of the global
batch, identical parameters on both ranks after the optimizer step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.Tanh(), torch.nn.Linear(32, 5))


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(4, 12, generator=g), torch.randn(4, 5, generator=g)


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ir2rgb_amd.vid2vid import FlatGrads
    m = _model()
    fg = FlatGrads(m.parameters(), chunk_elems=100)   # several chunks per buffer
    opt = torch.optim.Adam(fg.params, lr=1e-2, betas=(0.5, 0.999))
    x, y = _data(rank)
    for _ in range(3):
        fg.zero()
        torch.nn.functional.mse_loss(m(x), y).backward()
        fg.all_reduce_async(world)
        lo, hi = fg.flat.data_ptr(), fg.flat.data_ptr() + fg.flat.numel() * 4
        assert all(lo <= p.grad.data_ptr() < hi for p in fg.params)  # gathered: views of the flat buffer
        fg.wait()
        opt.step()
    # the flat buffer pads every view to a 16-byte boundary: compare the views, not the raw buffer
    torch.save((rank, torch.cat([v.reshape(-1) for v in fg.views]).clone(), torch.cat([p.detach().reshape(-1) for p in m.parameters()])),
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_matches_global_batch(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)  # raises with the child traceback
    res = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]
    # single-process reference: mean of the per-rank losses == loss over the global batch
    m = _model()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, betas=(0.5, 0.999))
    data = [_data(r) for r in range(world)]
    for _ in range(3):
        opt.zero_grad()
        sum(torch.nn.functional.mse_loss(m(x), y) for x, y in data).div(world).backward()
        last = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
        opt.step()
    ref_params = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    for rank, flat, params in res:
        torch.testing.assert_close(flat, last, atol=1e-6, rtol=1e-5)
        torch.testing.assert_close(params, ref_params, atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(res[0][2], res[1][2], atol=0, rtol=0)  # ranks stay bit-identical


class _SinkConv(torch.autograd.Function):
    """conv2d whose weight gradient goes where ir2rgb_amd.autograd.ConvStageFn puts it: into the parameter's sink slice of
    the flat all-reduce buffer when one is registered (autograd.GRAD_SINKS), returned as the gradient."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.conv2d(x, w, padding=1)

    @staticmethod
    def backward(ctx, g):
        from ir2rgb_amd import autograd as A
        x, w = ctx.saved_tensors
        dw = torch.nn.grad.conv2d_weight(x, w.shape, g, padding=1)
        sink = A.GRAD_SINKS.get(w) if A.GRAD_SINKS else None
        if sink is not None:
            sink.copy_(dw)
            dw = sink.view(w.shape)
        return torch.nn.grad.conv2d_input(x.shape, w, g, padding=1), dw


class _TwoUse(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.w1 = torch.nn.Parameter(torch.randn(4, 4, 3, 3) * 0.2)
        self.w2 = torch.nn.Parameter(torch.randn(4, 4, 3, 3) * 0.2)
        self.b = torch.nn.Parameter(torch.zeros(4))

    def forward(self, x, uses):
        h = x
        for _ in range(uses):       # uses = 2: what a window that generates two frames does to every generator weight
            h = torch.tanh(_SinkConv.apply(_SinkConv.apply(h, self.w1), self.w2) + self.b.view(1, -1, 1, 1))
        return h


def _sink_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ir2rgb_amd.vid2vid import FlatGrads
    m = _TwoUse()
    fg = FlatGrads(m.parameters(), chunk_elems=100, world=world, direct=True)
    g = torch.Generator().manual_seed(50 + rank)
    x = torch.randn(2, 4, 6, 6, generator=g)
    out = {}
    for uses in (1, 2, 1):
        fg.set_direct(uses == 1)          # (Vid2VidTrainer.generate: in-place sinks only for one use per pass)
        fg.zero()
        m(x, uses).square().mean().backward()
        fg.all_reduce_async(world)
        fg.wait()
        out[len(out)] = (uses, torch.cat([v.reshape(-1) for v in fg.views]).clone())
    torch.save((rank, x, out), os.path.join(outdir, f"sink{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_in_place_gradient_sinks_are_disarmed_for_multi_use_windows(tmp_path):
    """FlatGrads(direct=True): weight gradients written straight into the all-reduce buffer are only sound for ONE use
    of a parameter per backward pass.  A pass that uses every weight twice (n_frames_load = 2) must fall back to the
    gathered form (set_direct(False)) and still average to the global-batch gradient; the in-place form resumes after."""
    world, port = 2, _free_port()
    mp.spawn(_sink_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), f"sink{r}.pt")) for r in range(world)]
    m = _TwoUse()
    for step in range(3):
        uses = res[0][2][step][0]
        grads = []
        for _, x, _ in res:
            m.zero_grad()
            h = x
            for _ in range(uses):
                h = torch.tanh(torch.nn.functional.conv2d(torch.nn.functional.conv2d(h, m.w1, padding=1), m.w2, padding=1)
                               + m.b.view(1, -1, 1, 1))
            h.square().mean().backward()
            grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]))
        want = sum(grads) / world
        for rank, _, out in res:
            torch.testing.assert_close(out[step][1], want, atol=1e-6, rtol=1e-5)


def test_synthetic_sequences_differ_per_rank_and_are_deterministic():
    from ir2rgb_amd.vid2vid import synthetic_sequence
    a0, b0 = synthetic_sequence(4, 32, 48, 1234, "cpu")
    a0b, _ = synthetic_sequence(4, 32, 48, 1234, "cpu")
    a1, _ = synthetic_sequence(4, 32, 48, 2234, "cpu")
    assert a0.shape == (1, 4, 3, 32, 48) and b0.shape == a0.shape
    assert torch.equal(a0, a0b) and not torch.equal(a0, a1)
    assert a0.abs().max() <= 1.0
