"""Grouped scalar losses of the vid2vid loop body (libir2rgb_hip.so: losses.hip).

``fused_losses(terms, nslots, dtype)`` evaluates up to 16 terms in one launch (+ a tiny finish kernel)
and returns an fp32 tensor ``[nslots]``; each term adds ``weight * mean(...)`` into its slot.  Terms:

    ("l1",  a, b, weight, slot)            mean |a - b|        half tensors of identical dense layout; b is a constant
    ("mse", a, target, weight, slot)       mean (a - target)^2 fp32 tensor against a scalar (least-squares GAN loss)
    ("ml1", a, b, mask, weight, slot)      mean |a*m - b*m|    fp32 NCHW, mask [N,1,H,W]; b a constant tensor or None

Gradients flow to every ``a`` that requires them (one launch for the whole group).  These are the
reference's criterionFeat / criterionGAN / criterionFlow terms (discriminator.py:154-210, loss.py).
"""
import ctypes

import torch
from torch.autograd import Function

from . import _lib

_DT = {torch.bfloat16: 1, torch.float16: 2}
_KIND = {"l1": 0, "mse": 1, "ml1": 2}


def _dense_like(a, b):
    return a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype


def _build_items(terms, tensors, grads):
    """ctypes item array for ``terms`` (structure without tensors) + ``tensors`` (flat list) [+ grads]."""
    arr = (_lib.LossItem * len(terms))()
    ti = 0
    for i, t in enumerate(terms):
        it = arr[i]
        kind = t[0]
        it.kind = _KIND[kind]
        a = tensors[ti]
        ti += 1
        it.a = a.data_ptr()
        it.n = a.numel()
        it.ga = grads[i].data_ptr() if grads is not None and grads[i] is not None else None
        it.b = it.mask = None
        it.hw = it.chw = 0
        it.target = 0.0
        if kind == "l1":
            it.b = tensors[ti].data_ptr()
            ti += 1
            it.weight, it.slot = t[1], t[2]
        elif kind == "mse":
            it.target, it.weight, it.slot = t[1], t[2], t[3]
        else:
            has_b = t[1]
            if has_b:
                it.b = tensors[ti].data_ptr()
                ti += 1
            it.mask = tensors[ti].data_ptr()
            ti += 1
            it.hw = a.shape[2] * a.shape[3]
            it.chw = a.shape[1] * it.hw
            it.weight, it.slot = t[2], t[3]
    return arr


# Gradient destinations: (data_ptr, shape, dtype) of a loss input -> the tensor its gradient is to be written into
# (vid2vid.split_groups registers the pieces of a batched discriminator output with their slices of ONE gradient buffer,
# so the pieces' gradients need no gather copy).  The destination itself is returned as the gradient.
GRAD_DST = {}


def _dst_key(t):
    return (t.data_ptr(), tuple(t.shape), t.dtype)


# A destination is written ONCE per backward pass: a second term that differentiates the same piece in the same pass (the
# real pair's logits enter both compute_loss_D calls of a window, discriminator.py:134,:143) gets a buffer of its own and
# autograd adds the two -- two writes into one destination would make the engine sum two aliases of the LAST write.
_DST_USED = set()


def _take_dst(t):
    key = _dst_key(t)
    g = GRAD_DST.get(key)
    if g is None or key in _DST_USED:
        return None
    if not _DST_USED:
        torch.autograd.Variable._execution_engine.queue_callback(_DST_USED.clear)     # (when this backward pass ends)
    _DST_USED.add(key)
    return g


class _FusedLossFn(Function):
    @staticmethod
    def forward(ctx, terms, nslots, dt, *tensors):
        lib = _lib.lib()
        dev = tensors[0].device
        out = torch.zeros(nslots, dtype=torch.float32, device=dev)
        partial = torch.empty(lib.ir2rgb_loss_partial_elems(), dtype=torch.float32, device=dev)
        arr = _build_items(terms, tensors, None)
        with _lib.on_device(tensors[0]):
            rc = lib.ir2rgb_loss_multi_fwd(arr, len(terms), dt, partial.data_ptr(),
                                           out.data_ptr(), _lib.current_stream(tensors[0]))
        _lib.check(rc, "loss_multi_fwd")
        ctx.terms, ctx.dt = terms, dt
        ctx.save_for_backward(*tensors)
        return out

    @staticmethod
    def backward(ctx, gout):
        tensors = ctx.saved_tensors
        terms = ctx.terms
        # position of every term's `a` in the flat tensor list
        pos, ti = [], 0
        for t in terms:
            pos.append(ti)
            ti += 1 + (1 if t[0] == "l1" else 0) + ((1 if t[1] else 0) + 1 if t[0] == "ml1" else 0)
        grads = []
        for p in pos:
            g = None
            if ctx.needs_input_grad[3 + p]:
                g = _take_dst(tensors[p]) if GRAD_DST else None
                if g is None or g.stride() != tensors[p].stride():
                    g = torch.empty_like(tensors[p])
            grads.append(g)
        if any(g is not None for g in grads):
            gout = gout.contiguous().float()
            arr = _build_items(terms, tensors, grads)
            with _lib.on_device(gout):
                rc = _lib.lib().ir2rgb_loss_multi_bwd(arr, len(terms), ctx.dt, gout.data_ptr(),
                                                      _lib.current_stream(gout))
            _lib.check(rc, "loss_multi_bwd")
        res = [None] * len(tensors)
        for p, g in zip(pos, grads):
            res[p] = g
        return (None, None, None) + tuple(res)


def fused_losses(terms, nslots, dtype=torch.bfloat16):
    """See the module docstring.  Returns an fp32 tensor [nslots]."""
    if not 1 <= len(terms) <= _lib.LOSS_MAX_ITEMS:
        raise ValueError(f"fused_losses: 1..{_lib.LOSS_MAX_ITEMS} terms per group")
    spec, tensors = [], []
    for t in terms:
        kind, a = t[0], t[1]
        if not a.is_cuda:
            raise ValueError("fused_losses: GPU tensors only (no CPU fallback)")
        if kind == "l1":
            _, a, b, w, slot = t
            if a.dtype not in _DT or _DT[a.dtype] != _DT[dtype] or a.numel() % 8:
                raise ValueError("fused_losses l1: half tensors of the group's dtype, numel % 8 == 0")
            if not (a.is_contiguous() or a.is_contiguous(memory_format=torch.channels_last)):
                a = a.contiguous(memory_format=torch.channels_last)
            b = b.detach()
            if not _dense_like(a, b):
                b = b.to(a.dtype).contiguous(memory_format=torch.channels_last if not a.is_contiguous() else torch.contiguous_format)
                if not _dense_like(a, b):
                    raise ValueError("fused_losses l1: a and b need the same dense layout")
            spec.append(("l1", float(w), int(slot)))
            tensors += [a, b]
        elif kind == "mse":
            _, a, target, w, slot = t
            a = a.float().contiguous()
            spec.append(("mse", float(target), float(w), int(slot)))
            tensors.append(a)
        elif kind == "ml1":
            _, a, b, mask, w, slot = t
            a = a.float().contiguous()
            if a.dim() != 4 or mask.shape != (a.shape[0], 1, a.shape[2], a.shape[3]):
                raise ValueError("fused_losses ml1: a [N,C,H,W] and mask [N,1,H,W]")
            spec.append(("ml1", b is not None, float(w), int(slot)))
            tensors.append(a)
            if b is not None:
                if b.shape != a.shape:
                    raise ValueError("fused_losses ml1: a and b shapes differ")
                tensors.append(b.detach().float().contiguous())
            tensors.append(mask.detach().float().contiguous())
        else:
            raise ValueError(f"fused_losses: unknown term {kind!r}")
    if any(s < 0 or s >= nslots for s in (sp[-1] for sp in spec)) or nslots > 4:
        raise ValueError("fused_losses: slots must lie in [0, nslots), nslots <= 4")
    return _FusedLossFn.apply(tuple(spec), nslots, _DT[dtype], *tensors)
