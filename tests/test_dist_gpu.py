"""Two ranks on ONE GPU (gloo carrying CUDA tensors through the host): the trainer's N > 1 path end to end --
gradients written straight into the flat all-reduce buffers by the convolutions (generators), gathered by one
multi-tensor copy (discriminators), chunked asynchronous all-reduce overlapped with the next backward pass, wait before
each Adam step.  What must hold: the averaged gradient of every parameter equals the mean of the two ranks' local
gradients (each rank also runs a world-size-1 twin on its own window to obtain them), and both ranks hold identical
parameters after the step.  RCCL itself (backend "nccl", ReduceOp.AVG over xGMI) cannot be exercised on a one-GPU box and
is UNMEASURED; everything around the collective is what this covers."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from window_stub import stub_flow_and_conf
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ir2rgb_amd import vid2vid as V
    dev = torch.device("cuda:0")
    # (small chunks: the generators' gradient buffer then leaves in many pieces, most of them from the autograd hooks
    # while the backward pass is still running)
    kw = dict(seed=0, first_layer_gen_filters=64, compute_dtype=torch.float16, build_flow_net=False,
              allreduce_chunk_elems=1 << 20)
    tr, twin = V.Vid2VidTrainer(dev, world_size=world, **kw), V.Vid2VidTrainer(dev, world_size=1, **kw)
    tr.flow_net = twin.flow_net = stub_flow_and_conf
    copied = []
    orig_copy = torch._foreach_copy_
    torch._foreach_copy_ = lambda dst, src, *a, **k: (copied.append(sum(t.numel() for t in dst)), orig_copy(dst, src, *a, **k))[1]
    A, B = V.synthetic_sequence(3, 64, 128, 100 + rank, dev)
    twin.train_window(A, B)                                  # local gradients (Adam does not touch .grad)
    early = []
    orig_reduce = V.FlatGrads._reduce
    V.FlatGrads._reduce = lambda self, lo, hi: (early.append((bool(self._pending), hi - lo)), orig_reduce(self, lo, hi))[1]
    tr.train_window(A, B)
    V.FlatGrads._reduce = orig_reduce
    torch._foreach_copy_ = orig_copy
    res = {}
    for name, mods_a, mods_b in (("G", tr.netG, twin.netG), ("D", [tr.netD], [twin.netD])):
        avg = torch.cat([p.grad.reshape(-1) for m in mods_a for p in m.parameters()])
        local = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for m in mods_b for p in m.parameters()])
        both = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        mean = sum(both) / world
        res[name] = ((avg - mean).norm() / mean.norm()).item()
        params = torch.cat([p.detach().reshape(-1) for m in mods_a for p in m.parameters()])
        gathered = [torch.empty_like(params) for _ in range(world)]
        dist.all_gather(gathered, params)
        res[name + "_params_equal"] = bool(torch.equal(gathered[0], gathered[1]))
        res[name + "_local_differs"] = ((both[0] - both[1]).norm() / mean.norm()).item()
    n_g = sum(p.numel() for m in tr.netG for p in m.parameters())
    res["early_fraction_of_G"] = sum(n for armed, n in early if armed) / n_g   # sent from hooks, during the backward pass
    res["copied_fraction_of_G"] = copied[0] / n_g if copied else 0.0     # the first gather of the window is the generators'
    torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_window_on_one_gpu(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"))
        print("rank", r, res)
        assert res["G"] <= 1e-6 and res["D"] <= 1e-6, res             # averaged == mean of the local gradients
        assert res["G_params_equal"] and res["D_params_equal"], res   # ranks stay identical after the step
        assert res["G_local_differs"] > 1e-2                          # (the two ranks really saw different windows)
        assert res["copied_fraction_of_G"] < 0.02, res                # the generator's weight gradients were written in place
        assert res["early_fraction_of_G"] > 0.8, res                  # ... and left while the backward pass was running
