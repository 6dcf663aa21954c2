"""Adam over a parameter list as ONE launch per step (libir2rgb_hip.so: adam.hip).

Same update as ``torch.optim.Adam(params, lr, betas, eps=1e-8)`` with weight_decay 0 and amsgrad off,
which is what the reference builds (generator.py / discriminator.py ``torch.optim.Adam(params, lr=opt.lr,
betas=(opt.beta1, 0.999))``).  Moments live in two flat fp32 buffers; a device table of
{param, grad, exp_avg, exp_avg_sq, numel} rows is refreshed with the current gradient pointers before
every step (one small asynchronous copy from pinned memory).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import streamcheck as SC


class FusedAdam:
    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FusedAdam: no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise ValueError("FusedAdam: GPU parameters only (no CPU fallback)")
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise ValueError("FusedAdam: contiguous fp32 parameters on one device")
        # one parameter group, in torch.optim's shape: Model.update_learning_rate (base_model.py:103-108) walks
        # ``optimizer.param_groups`` and assigns ``param_group['lr']``; the step reads lr / betas / eps from here
        self.param_groups = [{"params": self.params, "lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0,
                              "amsgrad": False}]
        self.step_count = 0
        n = sum(p.numel() for p in self.params)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        lib = _lib.lib()
        chunk = lib.ir2rgb_adam_chunk_elems()
        rows = np.zeros((len(self.params), 5), dtype=np.int64)
        blocks, off = [], 0
        for i, p in enumerate(self.params):
            k = p.numel()
            rows[i] = (p.data_ptr(), 0, self.exp_avg.data_ptr() + 4 * off, self.exp_avg_sq.data_ptr() + 4 * off, k)
            blocks += [(i, c) for c in range((k + chunk - 1) // chunk)]
            off += k
        # two pinned staging copies used alternately: the host may run a step ahead of the GPU, and a
        # staging buffer is rewritten only after the asynchronous copy that last read it has completed
        self._rows_host = [torch.from_numpy(rows.copy()).pin_memory() for _ in range(2)]
        self._rows_np = [t.numpy() for t in self._rows_host]
        self._copied = [None, None]
        self._rows_dev = torch.empty_like(self._rows_host[0], device=dev)
        self._blocks = torch.tensor(blocks, dtype=torch.int32, device=dev)
        self._ptrs = [p.data_ptr() for p in self.params]
        self.device = dev

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):
        self.param_groups[0]["lr"] = value

    @property
    def betas(self):
        return self.param_groups[0]["betas"]

    @property
    def eps(self):
        return self.param_groups[0]["eps"]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def state_dict(self):
        """torch.optim.Adam's layout ({'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]}), so a
        checkpoint written here loads into torch.optim.Adam over the same parameter list and vice versa."""
        state = {}
        if self.step_count:
            for i in range(len(self.params)):
                m, v = self.moments(i)
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        if len(g["params"]) != len(self.params):
            raise ValueError("FusedAdam.load_state_dict: parameter count differs")
        self.param_groups[0].update({k: v for k, v in g.items() if k in ("lr", "betas", "eps")})
        self.param_groups[0]["betas"] = tuple(self.param_groups[0]["betas"])
        steps = {int(s["step"]) for s in sd["state"].values()}
        if len(steps) > 1:
            raise ValueError("FusedAdam.load_state_dict: parameters with different step counts (one launch updates all)")
        self.step_count = steps.pop() if steps else 0
        with torch.no_grad():
            if not sd["state"]:         # a state saved before the first step: start from zero moments, as torch.optim.Adam does
                self.exp_avg.zero_()
                self.exp_avg_sq.zero_()
            for i, s in sd["state"].items():
                m, v = self.moments(int(i))
                m.copy_(s["exp_avg"])
                v.copy_(s["exp_avg_sq"])

    def moments(self, i):
        """(exp_avg, exp_avg_sq) views of parameter i (for tests / checkpoints)."""
        off = sum(p.numel() for p in self.params[:i])
        k = self.params[i].numel()
        return self.exp_avg[off:off + k].view_as(self.params[i]), self.exp_avg_sq[off:off + k].view_as(self.params[i])

    @torch.no_grad()
    def step(self):
        k = self.step_count & 1
        if self._copied[k] is not None:
            self._copied[k].synchronize()
        rows = self._rows_np[k]
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                raise RuntimeError("FusedAdam.step: a parameter has no gradient (FlatGrads assigns zeros to unused ones)")
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = p.grad = g.float().contiguous()
            if p.data_ptr() != self._ptrs[i]:
                raise RuntimeError("FusedAdam.step: parameter storage moved since construction")
            rows[i, 1] = g.data_ptr()
        self._rows_dev.copy_(self._rows_host[k], non_blocking=True)
        if self._copied[k] is None:
            self._copied[k] = torch.cuda.Event()
        self._copied[k].record()
        self.step_count += 1
        with _lib.on_device(self.exp_avg):
            rc = _lib.lib().ir2rgb_adam_step(self._rows_dev.data_ptr(), self._blocks.data_ptr(),
                                             self._blocks.shape[0], self.lr, self.betas[0], self.betas[1], self.eps,
                                             self.step_count, _lib.current_stream(self.exp_avg))
        _lib.check(rc, "adam_step")
        # the kernel wrote the parameters behind autograd's back: bump their version counters so that
        # caches keyed on them (the packed MFMA weights of ir2rgb_amd.layers) are refreshed
        torch.autograd.graph.increment_version(self.params)
        if SC.ENABLED:
            for p in self.params:
                SC.produced(p, "fp32 parameter (Adam step)")
