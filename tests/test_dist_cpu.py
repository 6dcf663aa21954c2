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


def test_synthetic_sequences_differ_per_rank_and_are_deterministic():
    from ir2rgb_amd.vid2vid import synthetic_sequence
    a0, b0 = synthetic_sequence(4, 32, 48, 1234, "cpu")
    a0b, _ = synthetic_sequence(4, 32, 48, 1234, "cpu")
    a1, _ = synthetic_sequence(4, 32, 48, 2234, "cpu")
    assert a0.shape == (1, 4, 3, 32, 48) and b0.shape == a0.shape
    assert torch.equal(a0, a0b) and not torch.equal(a0, a1)
    assert a0.abs().max() <= 1.0
