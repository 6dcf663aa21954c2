"""torch.autograd bindings of the fused HIP stages (what ``loss.backward()`` walks).

Each stage of ir2rgb_amd.layers gets a ``torch.autograd.Function`` whose backward is again a
short sequence of libir2rgb_hip.so calls:

    gz --bn_bwd--> gy (grad wrt conv output), dgamma, dbeta
    gy --conv2d_fwd with the adjoint geometry (+ fold_reflect / xexpand_bwd)--> dx
    (x, gy) --wgrad--> dW

Status of the pieces (see DESIGN.md "what is hand-written"): activation/BatchNorm backward, all
data gradients, weight gradients (MFMA, transposed LDS reads), reflection fold and x-im2col adjoint
are HIP, as are the backward of the separable head convolutions and of the warp-blend.  No torch
convolution is left on the generator / discriminator path, forward or backward.
"""
import contextlib
import ctypes
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib
from . import conv as C
from . import layers as L
from . import streamcheck as SC
from . import stageplan

_DT = {torch.bfloat16: 1, torch.float16: 2}
_ADJ_DESCS = {}      # adjoint descriptors of strided convolutions (conv_dgrad), one object per geometry


def _p(t):

    # (a plain int: accepted by the fastcall bindings and by ctypes' c_void_p parameters alike; a c_void_p object per
    # argument cost 0.75 us, ten of them per launch)
    return t.data_ptr() if t is not None else 0


def _as_half_nhwc(g, dtype):
    if g.dtype == dtype and C.is_nhwc(g):
        return g
    if g.dtype == torch.float32:
        return L.to_nhwc_half(g, dtype)
    return g.to(dtype).contiguous(memory_format=torch.channels_last)


def bn_bwd(gz, y, scale, shift, mean, invstd, act, out=None, params=None):
    """-> (gy, dgamma, dbeta).  scale None: activation-only stage (dbeta is then the bias gradient).  ``out``: where gy
    goes (a sample-group slice of the batch's gradient tensor, see ConvStageFn); ``params`` = (dgamma, dbeta) of an earlier
    group of the same layer: this group's are added to them in the kernel (act | 32)."""
    n, ch, h, w = y.shape
    npix = n * h * w
    lib = _lib.lib()
    nblk = lib.ir2rgb_bn_bwd_blocks(npix, ch)
    if nblk < 0:
        _lib.check(nblk, "bn_bwd_blocks")
    dev = y.device
    if params is not None:
        dgamma, dbeta = params
        act |= 32
        buf = torch.empty((nblk * 2 + 3) * ch, dtype=torch.float32, device=dev)
        ppartial = buf.data_ptr()
    else:
        # one allocation: [dgamma | dbeta | the kernel's partial rows] (the rows only ever exist as an address)
        buf = torch.empty((nblk * 2 + 5) * ch, dtype=torch.float32, device=dev)
        dgamma, dbeta = buf[:ch], buf[ch:2 * ch]
        ppartial = buf.data_ptr() + 8 * ch
    gy = out if out is not None else torch.empty_like(y, memory_format=torch.channels_last)
    with _lib.on_device(y):
        rc = lib.ir2rgb_bn_bwd(gz, y, scale, shift, mean, invstd, gy, dgamma, dbeta,
                               ppartial, npix, ch, act, _DT[y.dtype], _lib.current_stream(y))
    _lib.check(rc, "bn_bwd")
    return gy, dgamma, dbeta


def thin_grad_expand(gz, dtype):
    """fp32 gradient [N,Cout<=8,H,W] of a thin output -> (g64, g8, dbias): 64- and 8-channel zero-padded
    channels_last half copies and the per-channel sum, one launch."""
    gz = gz.float().contiguous()
    n, cout, h, w = gz.shape
    g64 = C.empty_nhwc(n, 64, h, w, dtype, gz.device)
    g8 = C.empty_nhwc(n, 8, h, w, dtype, gz.device)
    dbias = torch.empty(cout, dtype=torch.float32, device=gz.device)
    with _lib.on_device(gz):
        rc = _lib.lib().ir2rgb_thin_grad_expand(gz, g64, g8, dbias, n, cout, h, w, _DT[dtype],
                                                _lib.current_stream(gz))
    _lib.check(rc, "thin_grad_expand")
    return g64, g8, dbias


def fold_reflect(dxpad, pad_h, pad_w=None):
    pad_w = pad_h if pad_w is None else pad_w
    n, ch, hp, wp = dxpad.shape
    h, w = hp - 2 * pad_h, wp - 2 * pad_w
    dx = C.empty_nhwc(n, ch, h, w, dxpad.dtype, dxpad.device)
    with _lib.on_device(dxpad):
        rc = _lib.lib().ir2rgb_fold_reflect(dxpad, dx, n, h, w, ch, pad_h, pad_w, _DT[dxpad.dtype],
                                            _lib.current_stream(dxpad))
    _lib.check(rc, "fold_reflect")
    return dx


def xexpand_bwd(dxe, cin, w, kw, stride_w, pad_w, pad_mode):
    n, _, h, wout = dxe.shape
    din = torch.empty((n, cin, h, w), dtype=torch.float32, device=dxe.device)
    with _lib.on_device(dxe):
        rc = _lib.lib().ir2rgb_xexpand_bwd(dxe, din, n, cin, h, w, wout, kw, stride_w, pad_w, pad_mode,
                                           _DT[dxe.dtype], _lib.current_stream(dxe))
    _lib.check(rc, "xexpand_bwd")
    return din


# ---------------------------------------------------------------------------------------------
# data gradient = a forward convolution with the adjoint geometry
# ---------------------------------------------------------------------------------------------
def _flip_swap(w):
    """Conv2d weight [Cout,Cin,kh,kw] -> adjoint Conv2d weight [Cin,Cout,kh,kw] (taps reversed)."""
    return w.permute(1, 0, 2, 3).flip(2, 3)


def _compose(f, g):
    if f is None:
        return g
    if g is None:
        return f
    return lambda w: g(f(w))


# ---------------------------------------------------------------------------------------------
# channel padding: any ngf / ndf behind the reference's factories (networks.py:51-82; generator.py:36 halves ngf per
# spatial scale).  The MFMA kernels want 64-multiples (BatchNorm backward: powers of two), so a layer whose width is not
# one runs at the next power of two >= 64: zero weight rows / columns, zero BatchNorm shift for the extra channels, which
# therefore stay exactly zero through convolution, BatchNorm (0 * scale + 0), ReLU and residual adds, forward and backward.
# Parameters and their gradients keep the reference's shapes.  Widths that are 64-multiples take none of this.
# ---------------------------------------------------------------------------------------------
def padded_width(c):
    return c if c % 64 == 0 and (c & (c - 1)) == 0 else max(64, 1 << (c - 1).bit_length())


def _pad_dim(t, dim, to, value=0.0):
    if t.shape[dim] == to:
        return t
    shape = list(t.shape)
    shape[dim] = to - t.shape[dim]
    return torch.cat([t, t.new_full(shape, value)], dim)


def _pad_weight_fn(cout_to, cin_to, transposed):
    """Conv2d weight [Cout,Cin,kh,kw] (ConvTranspose2d: [Cin,Cout,kh,kw]) -> zero-padded to the widths the kernels run at."""
    def f(w):
        a, b = (cin_to, cout_to) if transposed else (cout_to, cin_to)
        return _pad_dim(_pad_dim(w, 0, a), 1, b)
    return f


class PadChannelsFn(Function):
    """[N,C,H,W] half -> [N,Cp,H,W] channels_last half with zero extra channels (a tensor entering the padded domain from
    outside: coarse features, a stand-alone ResnetBlock call); the gradient is the slice."""

    @staticmethod
    def forward(ctx, x, cp):
        ctx.c = x.shape[1]
        out = torch.zeros((x.shape[0], cp, x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device).contiguous(
            memory_format=torch.channels_last)
        out[:, :ctx.c] = x
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.c], None


def pad_channels(x, cp):
    return x if x.shape[1] == cp else PadChannelsFn.apply(x, cp)


class _PaddedBN:
    """nn.BatchNorm2d seen at a padded width by bn_finalize: gamma 1 / beta 0 / running (0, 1) for the extra channels;
    ``commit`` writes the real channels' running statistics back."""

    def __init__(self, bn, cp):
        self.bn, self.num_features = bn, cp
        self.weight = _pad_dim(bn.weight.detach(), 0, cp, 1.0)
        self.bias = _pad_dim(bn.bias.detach(), 0, cp, 0.0)
        self.track_running_stats, self.momentum, self.eps = bn.track_running_stats, bn.momentum, bn.eps
        self.running_mean = None if bn.running_mean is None else _pad_dim(bn.running_mean, 0, cp, 0.0)
        self.running_var = None if bn.running_var is None else _pad_dim(bn.running_var, 0, cp, 1.0)
        self.num_batches_tracked = bn.num_batches_tracked

    def commit(self, training):
        if training and self.track_running_stats and self.running_mean is not None:
            c = self.bn.num_features
            self.bn.running_mean.copy_(self.running_mean[:c])
            self.bn.running_var.copy_(self.running_var[:c])


def conv_dgrad(gy, conv, spec, x_shape, weight_fn=None, tag="dgrad", out=None):
    """gy: grad wrt the convolution output (channels_last half, channels % 64 == 0).  ``weight_fn``
    maps conv.weight to the weight tensor the forward convolution actually used (x-expanded / padded
    forms).  Returns grad wrt the convolution input (channels_last half)."""
    kh, kw = spec["k"]
    (sh, sw), (ph, pw) = spec["stride"], spec["pad"]
    n, cin, hin, win = x_shape
    dt = gy.dtype
    if spec["transposed"]:
        # forward was ConvTranspose2d(W[cin][cout]); adjoint = Conv2d with the same memory as [out=cin][in=cout]
        desc = C.make_desc(tuple(gy.shape), cin, (kh, kw), (sh, sw), (ph, pw), C.PAD_ZERO, dt)
        wp = L.packed_weight(conv, desc, weight_fn, tag=tag)
        dx, _ = C.conv2d_fwd(desc, gy, wp, out=out)
        return dx
    if sh == 1 and sw == 1:
        if spec["pad_mode"] == C.PAD_REFLECT:
            if (kh, kw, ph, pw) == (3, 3, 1, 1):
                # the patch-staged kernel evaluates the adjoint of the reflection in place (border terms):
                # no padded 2-pixel-larger output grid, no fold pass
                dadj = C.make_desc(tuple(gy.shape), cin, 3, 1, 1, C.PAD_REFLECT_ADJ, dt)
                if C.kernel_name(dadj) == "conv3x3_patch_kernel":
                    dpack = C.make_desc(tuple(gy.shape), cin, 3, 1, 1, C.PAD_ZERO, dt)
                    wp = L.packed_weight(conv, dpack, weight_fn, tag=tag, adjoint=True)
                    dx, _ = C.conv2d_fwd(dadj, gy, wp, out=out)
                    return dx
            desc = C.make_desc(tuple(gy.shape), cin, (kh, kw), 1, (kh - 1, kw - 1), C.PAD_ZERO, dt)
            wp = L.packed_weight(conv, desc, weight_fn, tag=tag, adjoint=True)
            dxpad, _ = C.conv2d_fwd(desc, gy, wp)
            dx = fold_reflect(dxpad, ph, pw) if (ph or pw) else dxpad
            return dx if out is None else out.copy_(dx)
        desc = C.make_desc(tuple(gy.shape), cin, (kh, kw), 1, (kh - 1 - ph, kw - 1 - pw), C.PAD_ZERO, dt)
        wp = L.packed_weight(conv, desc, weight_fn, tag=tag, adjoint=True)
        dx, _ = C.conv2d_fwd(desc, gy, wp, out=out)
        return dx
    # strided zero-padded convolution: adjoint = transposed convolution reading W as [in=cout][out=cin]
    if spec["pad_mode"] != C.PAD_ZERO:
        raise NotImplementedError("data gradient of a strided reflect-padded convolution")
    hfull, wfull = (gy.shape[2] - 1) * sh - 2 * ph + kh, (gy.shape[3] - 1) * sw - 2 * pw + kw
    key = (n, gy.shape[2], gy.shape[3], gy.shape[1], hin, win, cin, kh, kw, sh, sw, ph, pw, C.PAD_ZERO, 1, _DT[dt], 0, 0, 0, 0, 0, 0)
    desc = _ADJ_DESCS.get(key)
    if desc is None:
        desc = _ADJ_DESCS[key] = C.sealed(C.ConvDesc(*key))
    assert 0 <= hin - hfull < sh and 0 <= win - wfull < sw, "adjoint geometry mismatch"
    wp = L.packed_weight(conv, desc, weight_fn, tag=tag)
    dx, _ = C.conv2d_fwd(desc, gy, wp, out=out)
    return dx


# ---------------------------------------------------------------------------------------------
# weight gradient: MFMA kernel (wgrad_mfma.hip).  Thin gradients (the 1-channel PatchGAN logits)
# are zero-padded to 8 channels so they take the same kernel.
# ---------------------------------------------------------------------------------------------
# Gradient sinks (data-parallel runs): weight parameter -> its slice of the optimizer's flat all-reduce buffer.  A
# convolution whose weight is registered here writes its weight gradient straight into that slice and returns it, so
# autograd adopts a tensor that already lives in the buffer and no gather copy precedes the all-reduce.  Only sound for
# a parameter that receives ONE contribution per backward pass (the caller's promise: vid2vid.FlatGrads(direct=True)).
GRAD_SINKS = {}


def conv_wgrad(x, gy, weight_shape, spec, out=None, accumulate=False):
    cin, cout = x.shape[1], gy.shape[1]
    if cin % 8:
        raise NotImplementedError("weight gradient needs an input channel count that is a multiple of 8")
    if cout % 8:
        if spec["transposed"]:
            raise NotImplementedError("thin transposed convolutions do not occur on this path")
        pad = (-cout) % 8
        gp = torch.zeros((gy.shape[0], cout + pad, gy.shape[2], gy.shape[3]), dtype=gy.dtype, device=gy.device).contiguous(
            memory_format=torch.channels_last)
        gp[:, :cout] = gy
        desc = C.make_desc(tuple(x.shape), cout + pad, spec["k"], spec["stride"], spec["pad"], spec["pad_mode"], x.dtype)
        return C.conv2d_wgrad(desc, x, gp)[:cout].contiguous()
    desc = C.make_desc(tuple(x.shape), cout, spec["k"], spec["stride"], spec["pad"], spec["pad_mode"], x.dtype,
                       bool(spec["transposed"]), spec.get("output_padding", 0))
    return C.conv2d_wgrad(desc, x, gy, out=out, accumulate=accumulate)


# ---------------------------------------------------------------------------------------------
# weight gradients on a second HIP stream (OFF by default since the nine-tap weight-gradient kernel: see below).
# The data-gradient chain (dgrad conv -> BN backward of the layer below) is the critical path of
# loss.backward(); the weight gradient of a layer only feeds the optimizer, so it can run beside that chain
# and fill the CUs the chain's small launches leave idle.  This is only sound for a parameter that is used
# ONCE per backward pass: autograd then hands the tensor to AccumulateGrad untouched.  A parameter used by
# several nodes has its contributions summed by the engine on the main stream, which knows nothing of the
# side stream (measured: discriminator gradients, 3 uses per pass, came out wrong).  So the side stream is
# opt-in per module tree (enable_side_wgrad: the trainer marks the generators, each called once per window)
# and never used while the parameter already holds a gradient.  The side stream is joined when the backward
# pass ends (engine callback).
# Measured: with the one-tap weight-gradient kernel (126 us, half of the chip idle) the overlap saved 2 ms per
# window; with the nine-tap kernel (44 us, every CU busy) it COSTS 1.7 ms (44.2 vs 42.5 ms per window): the
# overlapped kernels slow the critical chain and the stream hand-offs add bubbles -- also when only the
# remaining one-tap launches use the side stream (IR2RGB_WGRAD_STREAM=2: 44.3 ms).  Hence default off
# (IR2RGB_WGRAD_STREAM=1 enables it; tests/test_losses_gpu.py keeps it covered).
# ---------------------------------------------------------------------------------------------
WGRAD_SIDE_STREAM = os.environ.get("IR2RGB_WGRAD_STREAM", "0") != "0"
# "2": only weight gradients that do NOT fill the chip (strided / transposed / first layers, one-tap kernel) go to the
# side stream; the nine-tap kernel of the residual blocks occupies every CU and slows the data-gradient chain it overlaps
WGRAD_SIDE_SMALL_ONLY = os.environ.get("IR2RGB_WGRAD_STREAM", "0") == "2"
_SIDE = {}


def _join_side(dev_index):
    st = _SIDE[dev_index]
    st["pending"] = False
    C.SIDE_BUSY = False
    torch.cuda.current_stream(dev_index).wait_stream(st["stream"])


def enable_side_wgrad(module, enabled=True):
    """Allow (or forbid) side-stream weight gradients for every convolution under ``module``.  The caller
    asserts that each of these convolutions is applied once per backward pass."""
    for m in module.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            m._ir2rgb_side_wgrad = bool(enabled)


def wgrad_overlapped(conv, fn, *inputs):
    """Run ``fn()`` (the weight-gradient computation of ``conv`` reading ``inputs``) on the side stream
    when that is safe (see above), else on the current stream."""
    dev = inputs[0].device
    param = conv.weight
    if not WGRAD_SIDE_STREAM or dev.type != "cuda" or not getattr(conv, "_ir2rgb_side_wgrad", False):
        return fn()
    if GRAD_SINKS and param in GRAD_SINKS:
        # the gradient goes straight into the all-reduce buffer, and the hook that puts its chunk on the wire runs on the
        # main stream: it must not overtake a kernel on the side stream
        return fn()
    if WGRAD_SIDE_SMALL_ONLY and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1) and \
            not isinstance(conv, torch.nn.ConvTranspose2d) and conv.in_channels >= 64:
        return fn()
    st = _SIDE.get(dev.index)
    if st is None:
        st = _SIDE[dev.index] = {"stream": torch.cuda.Stream(dev), "pending": False}
    main, side = torch.cuda.current_stream(dev), st["stream"]
    if param.grad is not None:
        main.wait_stream(side)
        return fn()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        out = fn()
    for t in inputs:
        t.record_stream(side)
    out.record_stream(main)
    if not st["pending"]:
        st["pending"] = True
        C.SIDE_BUSY = True
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _join_side(dev.index))
    return out


# ---------------------------------------------------------------------------------------------
# backward flags.  A discriminator forward on generated frames serves two losses: the discriminator's
# (gradients for the discriminator parameters only) and the generator's (gradients for the frames
# only).  The reference runs that forward twice -- netD(fake.detach()) and netD(fake) with the result of
# the latter never reaching optimizer_D (discriminator.py:154-166) -- although both produce the same
# activations.  Here it runs once; which gradients a backward pass over it produces is selected by
# flags the trainer sets on the convolution modules around each loss.backward():
#   SKIP_PARAM_GRADS  no weight / bias / BatchNorm gradients (the generator's pass)
#   SKIP_INPUT_GRAD   an input layer ('first' stage) returns no gradient for the image (the
#                     discriminator's pass: nothing flows back into the generator)
# ---------------------------------------------------------------------------------------------
SKIP_PARAM_GRADS, SKIP_INPUT_GRAD = 1, 2
ACCUMULATE_IN_KERNEL = os.environ.get("IR2RGB_WGRAD_ACC", "1") != "0"    # dw += inside the weight-gradient kernel (see backward)
FUSED_BN = os.environ.get("IR2RGB_FUSED_BN", "1") != "0"    # bn_finalize + bn_apply in one launch where the statistics are few rows


@contextlib.contextmanager
def backward_flags(modules, flags, active_groups=None):
    """For the backward passes run inside: ``flags`` (SKIP_*) on every convolution stage of ``modules``; ``active_groups``
    = k: of the sample groups of a batched forward (conv_stage(groups=G)) only the first k receive a gradient in this
    pass -- the stages then work on that leading part of the batch only and leave the rest of every gradient tensor
    unwritten (nobody reads it: the generator's pass through the discriminators never reaches the real frames)."""
    convs = [m for mod in modules for m in mod.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))]
    for m in convs:
        m._ir2rgb_bwd = flags
        m._ir2rgb_active = active_groups
    try:
        yield
    finally:
        for m in convs:
            m._ir2rgb_bwd = 0
            m._ir2rgb_active = None


# ---------------------------------------------------------------------------------------------
# the fused stage
# ---------------------------------------------------------------------------------------------
def _fused_act(spec):
    """Activation fused into the convolution epilogue of a stage WITHOUT BatchNorm (ir2rgb_conv_desc.act)."""
    return 1 if spec["fused_leaky"] else (3 if spec.get("fused_relu") else 0)


class ConvStageFn(Function):
    """z = act(bn(conv(x))) + res1 + res2 on channels_last half tensors (x may be an NCHW fp32 image
    for 'first' stages).  Arguments after ``x``: weight, bias, gamma, beta (fp32 parameters), res1,
    res2, then the non-tensor ``spec`` dict and the conv / bn modules (packed-weight cache, BN
    buffers)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, res1, res2, spec, conv, bn):
        plan = stageplan.lookup(x, spec, conv, bn, FUSED_BN)
        if plan is not None:        # the plain stage, host work precomputed (ir2rgb_amd/stageplan.py): same launches
            return plan.forward(ctx, x, bias, res1, res2, conv, bn)
        ctx.plan = None
        dt = spec["dtype"]
        first = spec["first"]
        cout, cin = conv.out_channels, conv.in_channels
        cout_p = cout if spec.get("out_f32", False) else padded_width(cout)
        cin_p = cin if first else padded_width(cin)
        padded = cout_p != cout or cin_p != cin
        ctx.wfn = wfn = _pad_weight_fn(cout_p, cin_p, spec["transposed"]) if padded else None
        if first:
            kh, kw = conv.kernel_size
            xin = L.xexpand(x, kw, spec["stride"][1], spec["pad"][1], spec["pad_mode"], dt)
            desc = C.make_desc(tuple(xin.shape), cout_p, (kh, 1), (spec["stride"][0], 1), (spec["pad"][0], 0),
                               spec["pad_mode"], dt, act=_fused_act(spec))
            wp = L.packed_weight(conv, desc, _compose(wfn, L._xexpanded_weight(kw)), tag="xexp")
        else:
            if x.shape[1] != cin_p:
                raise ValueError(f"conv stage: input has {x.shape[1]} channels, the layer runs at {cin_p} (autograd.pad_channels)")
            xin = x
            desc = C.make_desc(tuple(x.shape), cout_p, spec["k"], spec["stride"], spec["pad"], spec["pad_mode"],
                               dt, spec["transposed"], spec.get("output_padding", 0), act=_fused_act(spec),
                               out_f32=spec.get("out_f32", False),
                               stats_per_sample=bn is not None and spec.get("groups", 1) > 1)
            wp = L.packed_weight(conv, desc, wfn, tag="wpad" if padded else "w")
        if padded and bias is not None:
            bias = _pad_dim(bias.detach(), 0, cout_p)
        scale = shift = mean = invstd = None
        ctx.frozen = False
        if bn is not None:
            # the bias of a convolution in front of BatchNorm cancels: it is left out of the activations and
            # handed to the statistics kernel, which needs it for the running mean only (ir2rgb_bn_finalize_ex)
            ctx.frozen = L.bn_frozen(bn, spec["training"])
            y, stats = C.conv2d_fwd(desc, xin, wp, None, want_stats=not ctx.frozen)
            bnp = _PaddedBN(bn, cout_p) if cout_p != cout else bn
            G = spec.get("groups", 1)
            count = desc.N * desc.Hout * desc.Wout
            fused = FUSED_BN and not ctx.frozen and spec["training"] and cout_p % 64 == 0
            if G == 1:
                if fused and stats.shape[0] <= L.FUSED_BN_MAX_ROWS:
                    # few partial rows (the residual blocks): statistics and apply in one launch
                    z, scale, shift, mean, invstd = L.bn_finalize_apply(stats, count, bnp, y, spec["act"], res1, res2, bias)
                else:
                    scale, shift, mean, invstd = L.bn_finalize(stats, count, bnp, spec["training"], bias)
                    z = L.bn_apply(y, scale, shift, spec["act"], res1, res2)
            else:
                # G independent forwards batched along N (the discriminators see real / generated / raw frames with the same
                # weights): ONE convolution, then BatchNorm per sample group exactly as G separate calls would run it --
                # statistics over the group's samples, running statistics advanced group by group
                if desc.N % G or res1 is not None or res2 is not None:
                    raise ValueError("conv stage: sample groups need N % groups == 0 and no residual inputs")
                ng, rg = desc.N // G, (0 if ctx.frozen else stats.shape[0] // G)
                z = torch.empty_like(y, memory_format=torch.channels_last)
                vec = torch.empty((4, G, cout_p), dtype=torch.float32, device=y.device)
                scale, shift, mean, invstd = vec[0], vec[1], vec[2], vec[3]
                reps = L._STAT_UPDATES        # layers.repeated_forward: an int, or one count per group
                if not ctx.frozen and bnp is bn and not SC.ENABLED and spec["training"]:
                    # the same per-group launches as the loop below, addressed by offset: no slices, no per-group wrappers
                    # (3 groups x 12 BatchNorm stages per window: ~7 tensor views and ~10 us of Python per group)
                    lib, stream = _lib.lib(), _lib.current_stream(y)
                    _, pw, pb, prm, prv, has_rm, momentum, eps, trs = L._bn_ptrs(bn)
                    track = trs and has_rm
                    py, pz, ps, pv = y.data_ptr(), z.data_ptr(), stats.data_ptr(), vec.data_ptr()
                    npix_g = count // G
                    per_y, per_s, c4 = npix_g * cout_p * y.element_size(), rg * 2 * cout_p * 4, cout_p * 4
                    gc4, act_, dtc = G * c4, spec["act"], _DT[y.dtype]
                    fuse_g = fused and rg <= L.FUSED_BN_MAX_ROWS
                    for g in (spec.get("group_order") or range(G)):
                        r = reps[g] if isinstance(reps, tuple) else reps
                        sc = pv + g * c4
                        if fuse_g:
                            rc = lib.ir2rgb_bn_finalize_apply(ps + g * per_s, rg, cout_p, npix_g, pw, pb, bias, prm if track else None,
                                                              prv if track else None, momentum, eps, sc, sc + gc4, sc + 2 * gc4,
                                                              sc + 3 * gc4, r, py + g * per_y, None, None, pz + g * per_y, npix_g,
                                                              act_, dtc, stream)
                            if rc:
                                _lib.check(rc, "bn_finalize_apply")
                        else:
                            rc = lib.ir2rgb_bn_finalize_ex(ps + g * per_s, rg, cout_p, npix_g, pw, pb, bias, prm if track else None,
                                                           prv if track else None, momentum, eps, sc, sc + gc4, sc + 2 * gc4,
                                                           sc + 3 * gc4, r, 0, stream)
                            if rc:
                                _lib.check(rc, "bn_finalize")
                            rc = lib.ir2rgb_bn_apply(py + g * per_y, sc, sc + gc4, None, None, pz + g * per_y, npix_g, cout_p, act_, dtc,
                                                     stream)
                            if rc:
                                _lib.check(rc, "bn_apply")
                        if track and bn.num_batches_tracked is not None:
                            L._PENDING_COUNTERS.append((bn.num_batches_tracked, r))
                    G = 0       # (done: the loop below is empty)
                try:
                    for g in ((spec.get("group_order") or range(G)) if G else ()):      # (the order the running statistics advance in)
                        L._STAT_UPDATES = reps[g] if isinstance(reps, tuple) else reps
                        yg, zg = y[g * ng:(g + 1) * ng], z[g * ng:(g + 1) * ng]
                        sg = None if ctx.frozen else stats[g * rg:(g + 1) * rg]
                        outs = (scale[g], shift[g], mean[g], invstd[g])
                        if fused and rg <= L.FUSED_BN_MAX_ROWS:
                            L.bn_finalize_apply(sg, count // G, bnp, yg, spec["act"], None, None, bias, out=zg, outs=outs)
                        else:
                            L.bn_finalize(sg, count // G, bnp, spec["training"], bias, outs=outs)
                            L.bn_apply(yg, scale[g], shift[g], spec["act"], out=zg)
                finally:
                    L._STAT_UPDATES = reps
            if bnp is not bn:
                bnp.commit(spec["training"])
        else:
            y, _ = C.conv2d_fwd(desc, xin, wp, bias)
            z = y
        ctx.spec, ctx.conv = spec, conv
        ctx.x_shape = tuple(x.shape)
        ctx.has_bn = bn is not None
        ctx.has_res = (res1 is not None, res2 is not None)
        ctx.save_for_backward(xin, y, scale, shift, mean, invstd)
        return z

    @staticmethod
    def backward(ctx, gz):
        plan = ctx.plan
        if plan is not None:
            conv = ctx.conv
            if not SC.ENABLED and (plan.groups > 1 or (not getattr(conv, "_ir2rgb_bwd", 0) and
                                                        getattr(conv, "_ir2rgb_active", None) is None)):
                return plan.backward(ctx, gz)       # (a grouped plan handles the backward flags itself)
            # backward flags on a planned stage (a discriminator run without sample groups): the general code below
            xin, y, vec = ctx.saved_tensors
            scale, shift, mean, invstd = vec.unbind(0)
            spec = plan.spec
            ctx.wfn, ctx.frozen, ctx.x_shape, ctx.has_bn = None, False, plan.x_shape, True
        else:
            spec, conv = ctx.spec, ctx.conv
            xin, y, scale, shift, mean, invstd = ctx.saved_tensors
        hdt = xin.dtype
        pad_fn = wfn = ctx.wfn
        cout, cin = conv.out_channels, conv.in_channels
        flags = getattr(conv, "_ir2rgb_bwd", 0)
        want_params = not (flags & SKIP_PARAM_GRADS)
        want_dx = ctx.needs_input_grad[0] and not (spec["first"] and (flags & SKIP_INPUT_GRAD))
        # sample groups of which only the leading k carry a gradient in this pass (backward_flags): work on that part
        G, k = spec.get("groups", 1), getattr(conv, "_ir2rgb_active", None)
        n_full = None
        if G > 1 and k is not None and k < G:
            if want_params:
                raise RuntimeError("conv stage: a pass with inactive sample groups cannot produce parameter gradients")
            n_full = y.shape[0]
            na = n_full // G * k
            gz, y, xin = gz[:na], y[:na], xin[:na]
            if scale is not None and scale.dim() == 2:
                scale, shift, mean, invstd = scale[:k], shift[:k], mean[:k], invstd[:k]
        if spec.get("out_f32", False):
            # thin fp32 output (PatchGAN logits): pad the gradient to 64 channels for the MFMA adjoint
            cout = y.shape[1]
            gy, gy_thin, dbias = thin_grad_expand(gz, hdt)      # 64-channel / 8-channel zero-padded halves, sum
            pad_fn = _compose(wfn, lambda w: torch.cat([w, w.new_zeros((64 - w.shape[0],) + tuple(w.shape[1:]))], 0))
            dgamma = dbeta = None
        elif ctx.has_bn:
            gz = _as_half_nhwc(gz, hdt)
            act = spec["act"] | (16 if ctx.frozen else 0)
            if scale.dim() == 1:
                gy, dgamma, dbeta = bn_bwd(gz, y, scale, shift, mean, invstd, act)
            elif not ctx.frozen and not SC.ENABLED and gz.is_contiguous(memory_format=torch.channels_last):
                # sample groups, addressed by offset (the launches of the loop in the next branch, without its slices and
                # per-group wrappers; one partial-row region serves all groups: same stream, one group after the other)
                G = scale.shape[0]
                ng, ch = y.shape[0] // G, y.shape[1]
                npix_g = ng * y.shape[2] * y.shape[3]
                lib, stream = _lib.lib(), _lib.current_stream(y)
                nblk = lib.ir2rgb_bn_bwd_blocks(npix_g, ch)
                if nblk < 0:
                    _lib.check(nblk, "bn_bwd_blocks")
                gy = torch.empty_like(y, memory_format=torch.channels_last)
                buf = torch.empty((nblk * 2 + 5) * ch, dtype=torch.float32, device=y.device)
                dgamma, dbeta = buf[:ch], buf[ch:2 * ch]
                pb_, c4 = buf.data_ptr(), ch * 4
                per = npix_g * ch * y.element_size()
                pg, py_, pgy = gz.data_ptr(), y.data_ptr(), gy.data_ptr()
                psc, psh, pmu, piv = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()
                dtc = _DT[y.dtype]
                for g in range(G):
                    rc = lib.ir2rgb_bn_bwd(pg + g * per, py_ + g * per, psc + g * c4, psh + g * c4, pmu + g * c4, piv + g * c4,
                                           pgy + g * per, pb_, pb_ + c4, pb_ + 2 * c4, npix_g, ch, act | (32 if g else 0), dtc, stream)
                    if rc:
                        _lib.check(rc, "bn_bwd")
            else:                       # sample groups: per-group BatchNorm backward into the batch's gradient tensor
                G = scale.shape[0]
                ng = y.shape[0] // G
                gy = torch.empty_like(y, memory_format=torch.channels_last)
                params = None
                for g in range(G):
                    sl = slice(g * ng, (g + 1) * ng)
                    _, dgamma, dbeta = bn_bwd(gz[sl], y[sl], scale[g], shift[g], mean[g], invstd[g], act, out=gy[sl], params=params)
                    params = (dgamma, dbeta)
                if ctx.frozen:
                    scale = scale[0]    # (evaluation mode: the same running statistics for every group)
            gy_thin = gy
            # training mode: BatchNorm removes the per-channel mean, the bias gradient is exactly 0 (None = zeros);
            # evaluation mode: the layer is affine in the bias, d/dbias = scale * sum g'
            dbias = dbeta * scale if ctx.frozen else None
        else:
            gz = _as_half_nhwc(gz, hdt)
            # LeakyReLU / ReLU keep the sign (ReLU: y > 0 <=> pre-activation > 0): mask from the stored output
            act = 2 if spec["fused_leaky"] else (1 if spec.get("fused_relu") else 0)
            gy, _, dbias = bn_bwd(gz, y, None, None, None, None, act)
            gy_thin = gy
            dgamma = dbeta = None
        dx = None
        if want_dx:
            if spec["first"]:
                kh, kw = conv.kernel_size
                sub = dict(spec, k=(kh, 1), stride=(spec["stride"][0], 1), pad=(spec["pad"][0], 0))
                if spec["pad_mode"] != C.PAD_ZERO:
                    raise NotImplementedError("input gradient of a reflect-padded first layer is never needed")
                # adjoint of the (kh x 1) convolution over the expanded image, then of the x-im2col
                n, _, h, wout = xin.shape
                dxe = conv_dgrad(gy, conv, sub, (n, 64, h, wout), _compose(wfn, L._xexpanded_weight(kw)), tag="dgrad_xexp")
                dx = xexpand_bwd(dxe, ctx.x_shape[1], ctx.x_shape[3], kw, spec["stride"][1], spec["pad"][1],
                                 spec["pad_mode"])
            elif n_full is not None:
                # the gradient tensor keeps the full batch's shape; the inactive groups' part stays unwritten
                dx = C.empty_nhwc(n_full, ctx.x_shape[1], ctx.x_shape[2], ctx.x_shape[3], hdt, gy.device)
                conv_dgrad(gy, conv, spec, (gy.shape[0],) + tuple(ctx.x_shape[1:]), pad_fn, out=dx[:gy.shape[0]])
            else:
                dx = conv_dgrad(gy, conv, spec, ctx.x_shape, pad_fn)
            if n_full is not None and dx.shape[0] != n_full:       # (first layer: NCHW fp32 image gradient)
                full = dx.new_empty((n_full,) + tuple(dx.shape[1:]))
                full[:dx.shape[0]].copy_(dx)
                dx = full
        gy = gy_thin
        dw = None
        if not want_params:
            dbias = dgamma = dbeta = None
        if ctx.needs_input_grad[1] and want_params:
            if spec["first"]:
                def first_wgrad():
                    kh, kw = conv.kernel_size
                    sub = dict(spec, k=(kh, 1), stride=(spec["stride"][0], 1), pad=(spec["pad"][0], 0))
                    gwe = conv_wgrad(xin, gy, (gy.shape[1], 64, kh, 1), sub)            # [co_p][64][kh][1]
                    gwe = gwe[:cout, :cin * kw, :, 0].reshape(cout, cin, kw, kh)        # [co][ci][kx][ky]
                    return gwe.permute(0, 1, 3, 2).contiguous()
                dw = wgrad_overlapped(conv, first_wgrad, xin, gy)
            else:
                if spec.get("out_f32", False):   # gy holds 8 zero-padded channels: the extra rows are dropped
                    wsh = (8,) + tuple(conv.weight.shape[1:])
                    dw = wgrad_overlapped(conv, lambda: conv_wgrad(xin, gy, wsh, spec)[:conv.out_channels].contiguous(),
                                          xin, gy)
                elif wfn is not None:   # the layer ran at padded widths: the parameter's gradient is the real corner
                    def padded_wgrad():
                        g = conv_wgrad(xin, gy, None, spec)
                        return (g[:cin, :cout] if spec["transposed"] else g[:cout, :cin]).contiguous()
                    dw = wgrad_overlapped(conv, padded_wgrad, xin, gy)
                else:
                    sink = GRAD_SINKS.get(conv.weight) if GRAD_SINKS else None
                    have = conv.weight.grad
                    if (ACCUMULATE_IN_KERNEL and sink is None and have is not None and have.dtype == torch.float32
                            and have.is_contiguous() and have.shape == conv.weight.shape):
                        # a later use of the same parameter in this pass (the discriminators see two or three inputs per
                        # window): dw is added to .grad by the kernel's own finish pass; autograd gets nothing to add
                        conv_wgrad(xin, gy, tuple(conv.weight.shape), spec, out=have, accumulate=True)
                    else:
                        dw = wgrad_overlapped(conv, lambda: conv_wgrad(xin, gy, tuple(conv.weight.shape), spec, out=sink), xin, gy)
        r1 = gz if ctx.has_res[0] else None
        r2 = gz if ctx.has_res[1] else None
        if wfn is not None:      # reference-shaped parameter gradients
            dbias = dbias[:cout] if dbias is not None else None
            dgamma = dgamma[:cout] if dgamma is not None else None
            dbeta = dbeta[:cout] if dbeta is not None else None
        return dx, dw, (dbias if ctx.needs_input_grad[2] else None), dgamma, dbeta, r1, r2, None, None, None


def conv_stage(x, conv, bn, act, pad_mode, dtype, *, first=False, stride=None, pad=None, transposed=False,
               output_padding=0, res1=None, res2=None, fused_leaky=False, training=True, out_f32=False, fused_relu=False,
               groups=1, group_order=None):
    """Autograd-aware stage: act(bn(conv(x))) + res1 + res2 (bn may be None).  ``groups`` > 1: the batch holds that many
    independent forwards (N / groups samples each) -- BatchNorm treats them as separate calls."""
    stride = tuple(conv.stride) if stride is None else C._pair(stride)
    pad = tuple(conv.padding) if pad is None else C._pair(pad)
    spec = dict(k=tuple(conv.kernel_size), stride=stride, pad=pad, pad_mode=pad_mode, transposed=transposed,
                output_padding=output_padding, act=act, fused_leaky=fused_leaky, fused_relu=fused_relu, training=training, dtype=dtype,
                first=first, out_f32=out_f32, groups=groups, group_order=group_order)
    gamma = bn.weight if bn is not None else None
    beta = bn.bias if bn is not None else None
    return ConvStageFn.apply(x, conv.weight, conv.bias, gamma, beta, res1, res2, spec, conv, bn)


# ---------------------------------------------------------------------------------------------
# heads and warp-blend: HIP forward; backward INTERIM through torch autograd recompute
# ---------------------------------------------------------------------------------------------
def _pad_rows(w, rows):
    return torch.cat([w, w.new_zeros((rows - w.shape[0],) + tuple(w.shape[1:]))], 0) if w.shape[0] < rows else w


class HeadFn(Function):
    """Separable 7x7 head(s) on one feature map: HIP forward (1x7 MFMA pass + head_finish) and HIP
    backward (head_finish_bwd -> adjoint 1x7 MFMA convolution + reflect fold for the feature gradient,
    MFMA wgrad for the kernels)."""

    @staticmethod
    def forward(ctx, feat, acts, mul, convs, *params):
        out = L.head_stage(feat, convs, acts, mul)
        ctx.acts, ctx.mul, ctx.convs = acts, mul, convs
        ctx.save_for_backward(feat, out)
        return out

    @staticmethod
    def backward(ctx, gout):
        feat, out = ctx.saved_tensors
        convs, acts, mul = ctx.convs, ctx.acts, ctx.mul
        kh, kw = convs[0].kernel_size
        n, cin, h, w = feat.shape
        cout = out.shape[1]
        CT = 64
        gout = gout.float().contiguous()
        dT = C.empty_nhwc(n, CT, h, w, feat.dtype, feat.device)
        dbias = torch.empty(cout, dtype=torch.float32, device=feat.device)
        rows = _lib.lib().ir2rgb_head_finish_bwd_rows(n, h, w)
        if rows < 0:
            _lib.check(rows, "head_finish_bwd_rows")
        partial = torch.empty((rows, 8), dtype=torch.float32, device=feat.device)
        packed_acts = 0
        for i, a in enumerate(acts):
            packed_acts |= (a & 15) << (4 * i)
        with _lib.on_device(feat):
            rc = _lib.lib().ir2rgb_head_finish_bwd(gout, out, dT, dbias, partial, n, h, w, cout, kh, CT, kh // 2,
                                                   packed_acts, float(mul), _DT[feat.dtype], _lib.current_stream(feat))
        _lib.check(rc, "head_finish_bwd")
        spec = dict(k=(1, kw), stride=(1, 1), pad=(0, kw // 2), pad_mode=C.PAD_REFLECT, transposed=False)
        gfeat = None
        if ctx.needs_input_grad[0]:
            holder = convs[0]
            key = ("ysplit_adj", _DT[feat.dtype], cin) + tuple((c.weight._version, c.weight.data_ptr()) for c in convs)
            cache = holder.__dict__.setdefault("_ir2rgb_packed", {})
            hit = cache.get("ysplit_adj")
            desc = C.make_desc((n, CT, h, w), cin, (1, kw), 1, (0, kw - 1), C.PAD_ZERO, feat.dtype)
            if hit is None or hit[0] != key:
                with torch.no_grad():
                    wcat = _pad_dim(torch.cat([c.weight.detach().float() for c in convs], 0), 1, cin)
                    wy = _pad_rows(L._ysplit_weight(wcat), CT).contiguous()      # [64][cin][1][kw] forward weight
                    cache["ysplit_adj"] = (key, C.pack_weight(desc, wy, adjoint=True))
                    if SC.ENABLED:
                        SC.produced(cache["ysplit_adj"][1], "packed head weight (adjoint)")
            if SC.ENABLED:
                SC.consumed(cache["ysplit_adj"][1], "packed head weight (adjoint)")
            dpad, _ = C.conv2d_fwd(desc, dT, cache["ysplit_adj"][1])
            gfeat = fold_reflect(dpad, 0, kw // 2)
        wdesc = C.make_desc(tuple(feat.shape), CT, (1, kw), 1, (0, kw // 2), C.PAD_REFLECT, feat.dtype)
        gw = C.conv2d_wgrad(wdesc, feat, dT)                                       # [64][cin][1][kw]
        gw = gw[:cout * kh, :, 0, :].reshape(cout, kh, cin, kw).permute(0, 2, 1, 3)                 # [cout][cin][kh][kw]
        gw = gw[:, :convs[0].in_channels].contiguous()          # (a padded feature map: the real input channels)
        gws, gbs, o = [], [], 0
        for c in convs:
            gws.append(gw[o:o + c.out_channels])
            gbs.append(dbias[o:o + c.out_channels])
            o += c.out_channels
        return (gfeat, None, None, None) + tuple(gws) + tuple(gbs)


def head_stage(feat, convs, acts, mul=1.0):
    params = [c.weight for c in convs] + [c.bias for c in convs]
    return HeadFn.apply(feat, list(acts), float(mul), list(convs), *params)


class WarpBlendFn(Function):
    """img_final = raw*w + warp(prev, flow)*(1-w): HIP forward and HIP backward w.r.t. raw, flow, w.
    A gradient w.r.t. prev (never needed on the training path: prev is detached, generator.py:153-154)
    falls back to torch autograd of the same formula."""

    @staticmethod
    def forward(ctx, raw, prev, flow, weight):
        raw, prev, flow, weight = raw.contiguous(), prev.contiguous(), flow.contiguous(), weight.contiguous()
        ctx.save_for_backward(raw, prev, flow, weight)
        return L.warp_blend(raw, prev, flow, weight)

    @staticmethod
    def backward(ctx, gout):
        raw, prev, flow, weight = ctx.saved_tensors
        gout = gout.float().contiguous()
        graw, gflow, gw = torch.empty_like(raw), torch.empty_like(flow), torch.empty_like(weight)
        n, _, h, w = raw.shape
        with _lib.on_device(raw):
            rc = _lib.lib().ir2rgb_warp_blend_bwd(gout, raw, prev, flow, weight, graw, gflow,
                                                  gw, n, prev.shape[1], h, w, _lib.current_stream(raw))
        _lib.check(rc, "warp_blend_bwd")
        gprev = None
        if ctx.needs_input_grad[1]:
            from .networks import get_grid
            with torch.enable_grad():
                p = prev.detach().requires_grad_()
                grid = get_grid(n, h, w, device=raw.device, dtype=flow.dtype)
                fln = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], 1)
                warp = F.grid_sample(p[:, -3:], (grid + fln).permute(0, 2, 3, 1), mode="bilinear", padding_mode="border",
                                     align_corners=False)
                gprev, = torch.autograd.grad(warp, p, gout * (1 - weight))
        return graw, gprev, gflow, gw


def warp_blend(raw, prev, flow, weight):
    return WarpBlendFn.apply(raw, prev, flow, weight)


class AvgPool3s2Fn(Function):
    """AvgPool2d(3, stride 2, padding 1, count_include_pad=False) of an fp32 [..., H, W] tensor (ir2rgb_avgpool3s2)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        h, w = x.shape[-2:]
        ctx.shape = tuple(x.shape)
        y = torch.empty(tuple(x.shape[:-2]) + ((h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=torch.float32, device=x.device)
        with _lib.on_device(x):
            rc = _lib.lib().ir2rgb_avgpool3s2(x, y, x.numel() // (h * w), h, w, 0, _lib.current_stream(x))
        _lib.check(rc, "avgpool3s2")
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        h, w = ctx.shape[-2:]
        with _lib.on_device(g):
            rc = _lib.lib().ir2rgb_avgpool3s2(g, gx, gx.numel() // (h * w), h, w, 1, _lib.current_stream(g))
        _lib.check(rc, "avgpool3s2 backward")
        return gx


def avg_pool3s2(x):
    """The pyramids' down-sampling step; torch's operator for anything but fp32 GPU tensors."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2:
        return AvgPool3s2Fn.apply(x)
    return torch.nn.functional.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)


class AddFn(Function):
    """a + b on channels_last half tensors (HIP), gradient passed to both."""

    @staticmethod
    def forward(ctx, a, b):
        return L.bn_apply_add(a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return AddFn.apply(a, b)


class ToHalfFn(Function):
    """NCHW fp32 (or any) -> channels_last half (HIP converter); backward converts the gradient back."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src_dtype = x.dtype
        return L.to_nhwc_half(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return (L.to_nchw_f32(g) if ctx.src_dtype == torch.float32 else g.to(ctx.src_dtype)), None


def to_nhwc_half(x, dtype):
    if x.dtype == dtype and C.is_nhwc(x):
        return x
    return ToHalfFn.apply(x, dtype)
