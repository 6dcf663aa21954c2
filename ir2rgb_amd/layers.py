"""Fused layer primitives on NHWC half tensors, each a thin sequence of libir2rgb_hip.so calls.

A "stage" is what the reference writes as ``[pad, conv, norm, activation]`` module runs
(models/networks.py:141-171, :253-271, :556-580, :678-699):

    conv (MFMA implicit GEMM, bias + per-tile BatchNorm partial sums fused in the epilogue)
      -> bn_finalize (batch statistics, running-stat update)
      -> bn_apply    (scale/shift + ReLU / LeakyReLU + up to two residual adds)

Parameters stay in the fp32 ``nn.Conv2d`` / ``nn.BatchNorm2d`` containers the state_dict
exposes; the half-precision packed copy the MFMA kernel streams is cached per module and
refreshed whenever the fp32 weight's version counter moves (optimizer step, load_state_dict).
"""
import contextlib
import ctypes

import torch
import torch.nn as nn

from . import _lib
from . import conv as C
from . import streamcheck as SC

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
_DT = {torch.bfloat16: 1, torch.float16: 2}


def _p(t):

    # (a plain int: accepted by the fastcall bindings and by ctypes' c_void_p parameters alike; a c_void_p object per
    # argument cost 0.75 us, ten of them per launch)
    return t.data_ptr() if t is not None else 0


def _require_gpu(t, what):
    if not t.is_cuda:
        raise ValueError(f"{what}: ir2rgb_amd runs on an AMD GPU only (no CPU fallback)")


# ---------------------------------------------------------------------------------------------
# packed-weight cache
# ---------------------------------------------------------------------------------------------
def _gather(src, idx, dst):
    if SC.ENABLED:
        SC.consumed(src, "fp32 parameter"), SC.consumed(idx, "gather plan")
    with _lib.on_device(src):
        rc = _lib.lib().ir2rgb_gather_f32(src, idx, dst, dst.numel(), _lib.current_stream(src))
    _lib.check(rc, "gather_f32")
    if SC.ENABLED:
        SC.produced(dst, "rearranged fp32 weight")
    return dst


def _gather_plan(mod, tag, weight, src):
    """Index map of the rearrangement ``weight`` (a pure data movement: permute / reshape / zero-pad / cat with zeros) of
    ``mod.weight``, found by running it once on a tensor of 1-based element numbers (0 = a zero the rearrangement put
    there); cached per (module, tag).  None when ``weight`` is not such a map (checked once against its own result)."""
    plans = mod.__dict__.setdefault("_ir2rgb_gather", {})
    key = (src.data_ptr(), tuple(src.shape))
    hit = plans.get(tag)
    if hit is not None and hit[0] == key:
        return hit[1]
    plan = None
    n = src.numel()
    if n < (1 << 24):                # element numbers stay exact in fp32
        with torch.no_grad():
            r = weight(torch.arange(1, n + 1, dtype=torch.float32, device=src.device).view_as(src)).float().contiguous()
            idx = (r.round().to(torch.int32) - 1).contiguous()
            wbuf = torch.empty(r.shape, dtype=torch.float32, device=src.device)
            ref = weight(src.detach()).float().contiguous()
            if ref.shape == wbuf.shape and torch.equal(_gather(src.detach().contiguous().view(-1), idx, wbuf), ref):
                plan = (idx, wbuf)
    plans[tag] = (key, plan)
    return plan


def packed_weight(mod, desc, weight=None, tag="w", adjoint=False):
    """Packed half copy of ``mod.weight`` (or of ``weight(mod.weight)``, a rearranged form) for ``desc``.
    Cache entry: (key, packed, descriptor copy | None, adjoint, fp32 source of the pack, gather plan | None)."""
    src = mod.weight
    key = (tag, desc.dtype, src._version, src.data_ptr(), desc.Cin, desc.Cout, desc.kh, desc.kw, desc.transposed)
    cache = mod.__dict__.setdefault("_ir2rgb_packed", {})
    hit = cache.get(tag)
    if hit is not None and hit[0] == key:
        if SC.ENABLED:
            SC.consumed(hit[1], f"packed weight '{tag}'")
        return hit[1]
    plan = None
    if SC.ENABLED:
        SC.consumed(src, "fp32 parameter")
    with torch.no_grad():
        if weight is None:
            w = src.detach()
        else:
            plan = _gather_plan(mod, tag, weight, src) if src.dtype == torch.float32 and src.is_contiguous() else None
            w = _gather(src.detach().view(-1), plan[0], plan[1]) if plan is not None else weight(src.detach())
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.float().contiguous()
        packed = C.pack_weight(desc, w, adjoint=adjoint)
    if SC.ENABLED:
        SC.produced(packed, f"packed weight '{tag}'")
    # an entry packed from the parameter's own storage, or from a persistent rearranged copy that ONE gather launch
    # refreshes, can be refreshed in place by WeightRepacker; anything else repacks lazily at its next use
    managed = (weight is None and w.data_ptr() == src.data_ptr()) or plan is not None
    cache[tag] = (key, packed, C.ConvDesc.from_buffer_copy(desc) if managed else None, bool(adjoint), w if managed else None, plan)
    global _PACK_EPOCH
    _PACK_EPOCH += 1            # (a new packed buffer exists: WeightRepacker re-collects its entries)
    return packed


_PACK_EPOCH = 0


class WeightRepacker:
    """Refreshes, with ONE launch, every managed packed-weight copy cached under ``modules`` (forward and
    data-gradient operands) -- call ``run()`` right after the optimizer step that changed the weights.
    Without it each layer repacks lazily at its next use, ~280 small launches per training window.
    The packed buffers are overwritten in place, so no autograd graph recorded before the step may be
    run backward after it (torch itself refuses that for in-place updated parameters).  Rearranged weights (the first
    layers' x-im2col form, zero-padded widths) are refreshed by one gather launch each into their persistent fp32 copy,
    which the batched pack then reads; entries whose rearrangement is not a pure index map keep their lazy path."""

    def __init__(self, modules):
        self.modules = [m for root in modules for m in root.modules() if getattr(m, "weight", None) is not None]
        self.batch, self.sig, self.entries = None, None, []

    def _collect(self):
        ent = []
        for m in self.modules:
            for tag, hit in m.__dict__.get("_ir2rgb_packed", {}).items():
                if len(hit) == 6 and hit[2] is not None and hit[0][3] == m.weight.data_ptr():
                    ent.append((m, tag, hit))
        return ent

    def run(self):
        # the entries (and the device job table) only change when some layer packed a NEW buffer since the last run (first
        # use, new shape, moved parameter storage): walking ~280 cache entries and hashing their addresses every window was
        # ~1 ms of host time
        if getattr(self, "_epoch", None) == _PACK_EPOCH and self.entries:
            ent = self.entries
            sig = self.sig
        else:
            ent = self._collect()
            self.entries, self._epoch = ent, _PACK_EPOCH
            sig = tuple((id(m), tag, hit[1].data_ptr(), hit[0][3], hit[0][1], hit[4].data_ptr()) for m, tag, hit in ent)
        if not ent:
            return 0
        if sig != self.sig:      # first call, or a layer packed a new buffer since (new shape / first backward)
            by_dtype = {}
            for e in ent:
                by_dtype.setdefault(e[2][0][1], []).append(e)
            self.batch = [C.PackBatch([(hit[2], hit[4], hit[1], hit[3]) for m, tag, hit in es])
                          for es in by_dtype.values()]
            self.sig = sig
        with torch.no_grad():
            for m, tag, hit in ent:
                if hit[5] is not None:
                    _gather(m.weight.detach().view(-1), hit[5][0], hit[5][1])
        if SC.ENABLED:
            for m, tag, hit in ent:
                SC.consumed(m.weight, "fp32 parameter")
        for b in self.batch:
            b.run()
        if SC.ENABLED:
            for m, tag, hit in ent:
                SC.produced(hit[1], f"packed weight '{tag}' (batched repack)")
        for m, tag, hit in ent:
            k = hit[0]
            m._ir2rgb_packed[tag] = ((k[0], k[1], m.weight._version) + tuple(k[3:]),) + tuple(hit[1:])
        return len(ent)


# ---------------------------------------------------------------------------------------------
# BatchNorm pieces
# ---------------------------------------------------------------------------------------------
# How many identical forwards of the reference the current forward stands for (running statistics and
# num_batches_tracked advance that many times); set with ``repeated_forward``.
_STAT_UPDATES = 1


@contextlib.contextmanager
def repeated_forward(times):
    """The forwards inside stand for ``times`` identical reference forwards each (their batch statistics enter the
    running statistics that often).  A tuple: one count per sample group of a batched forward (ConvStageFn)."""
    global _STAT_UPDATES
    old, _STAT_UPDATES = _STAT_UPDATES, (tuple(int(t) for t in times) if isinstance(times, (tuple, list)) else int(times))
    try:
        yield
    finally:
        _STAT_UPDATES = old


def _bn_ptrs(bn):
    """ctypes pointers of a BatchNorm module's tensors, cached on the module (nn.Module attribute look-ups and ctypes
    conversions cost ~10 us per call, twice per stage and pass); rebuilt when the weight's storage moves."""
    hit = bn.__dict__.get("_ir2rgb_ptrs")
    w = bn.weight
    if hit is not None and hit[0] == (w.data_ptr() if w is not None else 0):
        return hit
    rm, rv = bn.running_mean, bn.running_var
    hit = (w.data_ptr() if w is not None else 0, w, bn.bias, rm, rv, rm is not None,
           0.1 if bn.momentum is None else float(bn.momentum), float(bn.eps), bool(bn.track_running_stats))
    bn.__dict__["_ir2rgb_ptrs"] = hit
    return hit


def bn_frozen(bn, training):
    """True when nn.BatchNorm2d would normalise with its running statistics (module.eval() and tracked stats)."""
    return (not training) and bn.track_running_stats and bn.running_mean is not None


def bn_finalize(stats, count, bn, training=True, conv_bias=None, outs=None):
    """-> (scale, shift, mean, invstd) fp32 [C] for an input that does NOT carry ``conv_bias`` (the bias of the
    convolution in front; BatchNorm cancels it, see ir2rgb_bn_finalize_ex).  Training mode: batch statistics
    from the convolution's partial sums, running statistics updated like nn.BatchNorm2d.  Evaluation mode
    (``training=False`` with tracked statistics): the running statistics, nothing updated; ``stats`` may be None."""
    frozen = bn_frozen(bn, training)
    ch = bn.num_features
    dev = bn.weight.device if bn.weight is not None else stats.device
    rows = 0 if stats is None else stats.shape[0]
    if outs is not None:        # caller-owned fp32 [C] vectors (a sample group's rows of ConvStageFn's [4][G][C] block)
        scale, shift, mean, invstd = outs
    else:
        scale = torch.empty(ch, dtype=torch.float32, device=dev)
        shift = torch.empty_like(scale)
        mean = torch.empty_like(scale)
        invstd = torch.empty_like(scale)
    track = training and bn.track_running_stats and bn.running_mean is not None
    momentum = 0.1 if bn.momentum is None else bn.momentum
    use_running = track or frozen
    if SC.ENABLED and use_running:
        SC.consumed(bn.running_mean, "BatchNorm running statistics")
        if track:
            SC.produced(bn.running_mean, "BatchNorm running statistics")
    with _lib.on_device(scale):
        rc = _lib.lib().ir2rgb_bn_finalize_ex(stats, rows, ch, int(count), bn.weight, bn.bias, conv_bias,
                                              bn.running_mean if use_running else None,
                                              bn.running_var if use_running else None, float(momentum),
                                              float(bn.eps), scale, shift, mean, invstd, _STAT_UPDATES,
                                              int(frozen), _lib.current_stream(scale))
    _lib.check(rc, "bn_finalize")
    if track and bn.num_batches_tracked is not None:
        _PENDING_COUNTERS.append((bn.num_batches_tracked, _STAT_UPDATES))
    return scale, shift, mean, invstd


FUSED_BN_MAX_ROWS = 128      # IR2RGB_BN_FUSED_MAX_ROWS of include/ir2rgb_hip.h


def bn_finalize_apply(stats, count, bn, y, act, res1=None, res2=None, conv_bias=None, out=None, outs=None):
    """Training-mode ``bn_finalize`` + ``bn_apply`` as one launch (ir2rgb_bn_finalize_apply) for convolutions with few
    partial rows.  -> (z, scale, shift, mean, invstd), bit-identical to the two calls."""
    rows, _, ch = stats.shape
    n, _, h, w = y.shape
    if outs is not None:
        scale, shift, mean, invstd = outs
    else:
        vec = torch.empty((4, ch), dtype=torch.float32, device=y.device)   # scale | shift | mean | invstd: one allocation
        scale, shift, mean, invstd = vec[0], vec[1], vec[2], vec[3]
    z = out if out is not None else torch.empty_like(y, memory_format=torch.channels_last)
    if isinstance(bn, nn.Module):
        _, pw, pb, prm, prv, has_rm, momentum, eps, trs = _bn_ptrs(bn)
    else:       # a padded shadow (autograd._PaddedBN): fresh tensors every call
        pw, pb, prm, prv, has_rm = bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.running_mean is not None
        momentum, eps, trs = 0.1 if bn.momentum is None else float(bn.momentum), float(bn.eps), bn.track_running_stats
    track = trs and has_rm
    null = 0
    if SC.ENABLED and track:
        SC.consumed(bn.running_mean, "BatchNorm running statistics")
        SC.produced(bn.running_mean, "BatchNorm running statistics")
    with _lib.on_device(y):
        rc = _lib.lib().ir2rgb_bn_finalize_apply(stats, rows, ch, int(count), pw, pb, conv_bias,
                                                 prm if track else null, prv if track else null, momentum, eps,
                                                 scale, shift, mean, invstd,
                                                 _STAT_UPDATES, y, res1, res2, z, n * h * w, act,
                                                 _DT[y.dtype], _lib.current_stream(y))
    _lib.check(rc, "bn_finalize_apply")
    if track and bn.num_batches_tracked is not None:
        _PENDING_COUNTERS.append((bn.num_batches_tracked, _STAT_UPDATES))
    return z, scale, shift, mean, invstd


_PENDING_COUNTERS = []


def flush_bn_counters():
    """num_batches_tracked += 1 for every BatchNorm touched since the last flush -- one fused launch
    (called at the end of each generator / discriminator forward)."""
    if _PENDING_COUNTERS:
        total = {}                  # per counter tensor (a batched forward touches a layer once per sample group)
        for t, k in _PENDING_COUNTERS:
            ent = total.setdefault(id(t), [t, 0])
            ent[1] += k
        for k in sorted({k for _, k in total.values()}):
            torch._foreach_add_([t for t, kk in total.values() if kk == k], k)
        _PENDING_COUNTERS.clear()


def bn_apply(x, scale, shift, act=ACT_NONE, res1=None, res2=None, out=None):
    n, ch, h, w = x.shape
    y = out if out is not None else torch.empty_like(x, memory_format=torch.channels_last)
    with _lib.on_device(x):
        rc = _lib.lib().ir2rgb_bn_apply(x, scale, shift, res1, res2, y, n * h * w, ch, act,
                                        _DT[x.dtype], _lib.current_stream(x))
    _lib.check(rc, "bn_apply")
    return y


# ---------------------------------------------------------------------------------------------
# layout converters
# ---------------------------------------------------------------------------------------------
def to_nhwc_half(x, dtype):
    """NCHW fp32 -> logical NCHW / channels_last half (zero copy if already in that form)."""
    if x.dtype == dtype and C.is_nhwc(x):
        return x
    _require_gpu(x, "to_nhwc_half")
    x = x.float().contiguous()
    n, ch, h, w = x.shape
    y = C.empty_nhwc(n, ch, h, w, dtype, x.device)
    with _lib.on_device(x):
        rc = _lib.lib().ir2rgb_nchw_f32_to_nhwc_half(x, y, n, ch, h, w, _DT[dtype], _lib.current_stream(x))
    _lib.check(rc, "nchw_f32_to_nhwc_half")
    return y


def to_nchw_f32(x):
    """channels_last half -> contiguous NCHW fp32."""
    if x.dtype == torch.float32 and x.is_contiguous():
        return x
    if not C.is_nhwc(x) or x.dtype not in _DT:
        return x.float().contiguous()
    n, ch, h, w = x.shape
    y = torch.empty((n, ch, h, w), dtype=torch.float32, device=x.device)
    with _lib.on_device(x):
        rc = _lib.lib().ir2rgb_nhwc_half_to_nchw_f32(x, y, n, ch, h, w, _DT[x.dtype], _lib.current_stream(x))
    _lib.check(rc, "nhwc_half_to_nchw_f32")
    return y


def xexpand(x, kw, stride_w, pad_w, pad_mode, dtype, cx=64):
    """NCHW fp32 [N,Cin,H,W] -> channels_last half [N,cx,H,Wout] with channel ci*kw+kx = x[ci][.., ox*s+kx-p]."""
    _require_gpu(x, "xexpand")
    x = x.float().contiguous()
    n, cin, h, w = x.shape
    wout = (w + 2 * pad_w - kw) // stride_w + 1
    y = C.empty_nhwc(n, cx, h, wout, dtype, x.device)
    with _lib.on_device(x):
        rc = _lib.lib().ir2rgb_xexpand_cx(x, y, n, cin, h, w, wout, kw, stride_w, pad_w, pad_mode, cx, _DT[dtype],
                                          _lib.current_stream(x))
    _lib.check(rc, "xexpand")
    return y


def _xexpanded_weight(kw, cx=64):
    """[Cout,Cin,kh,kw] -> [Cout,cx,kh,1] with input channel ci*kw+kx (zero padded to cx)."""
    def f(w):
        co, ci, kh, kw_ = w.shape
        assert kw_ == kw and ci * kw <= cx
        v = w.permute(0, 2, 1, 3).reshape(co, kh, ci * kw)       # [co][ky][ci*kw+kx]
        v = torch.nn.functional.pad(v, (0, cx - ci * kw))        # [co][ky][cx]
        return v.permute(0, 2, 1).unsqueeze(-1)                  # [co][cx][ky][1]
    return f


def _ysplit_weight(w):
    """[Cout,Cin,kh,kw] -> [pad24(Cout*kh),Cin,1,kw]: output channel co*kh+ky holds row ky of the kernel."""
    co, ci, kh, kw = w.shape
    v = w.permute(0, 2, 1, 3).reshape(co * kh, ci, 1, kw)
    pad = (-v.shape[0]) % 8
    if pad:
        v = torch.cat([v, v.new_zeros(pad, ci, 1, kw)], 0)
    return v


# ---------------------------------------------------------------------------------------------
# stages
# ---------------------------------------------------------------------------------------------
def conv_stage(x, conv, bn, act, pad_mode, *, stride=None, pad=None, transposed=False, output_padding=0,
               res1=None, res2=None, fused_leaky=False, training=True):
    """x (channels_last half) -> act(bn(conv(x))) [+ res1 + res2].  ``bn`` may be None."""
    stride = conv.stride if stride is None else stride
    pad = conv.padding if pad is None else pad
    desc = C.make_desc(tuple(x.shape), conv.out_channels, conv.kernel_size, stride, pad, pad_mode, x.dtype, transposed,
                       output_padding, act=1 if fused_leaky else 0)
    wp = packed_weight(conv, desc)
    if bn is None:
        return C.conv2d_fwd(desc, x, wp, conv.bias)[0]
    y, stats = C.conv2d_fwd(desc, x, wp, None, want_stats=not bn_frozen(bn, training))
    scale, shift, _, _ = bn_finalize(stats, desc.N * desc.Hout * desc.Wout, bn, training, conv.bias)
    return bn_apply(y, scale, shift, act, res1, res2, out=y)


def first_stage(x_nchw, conv, bn, act, pad_mode, dtype, *, fused_leaky=False, training=True):
    """Small-Cin first layer on an NCHW fp32 image: x-im2col + (kh x 1) MFMA convolution.

    ReflectionPad2d(3)+Conv7x7 (networks.py:141,:150,:253-255) or Conv4x4 s2 p2 (:680)."""
    kh, kw = conv.kernel_size
    sh, sw = conv.stride
    ph, pw = (3, 3) if pad_mode == C.PAD_REFLECT else conv.padding
    if conv.in_channels * kw > 64:
        raise NotImplementedError(f"first layer with {conv.in_channels} input channels x kernel width {kw} > 64")
    xe = xexpand(x_nchw, kw, sw, pw, pad_mode, dtype)
    desc = C.make_desc(tuple(xe.shape), conv.out_channels, (kh, 1), (sh, 1), (ph, 0), pad_mode, dtype,
                       act=1 if fused_leaky else 0)
    wp = packed_weight(conv, desc, _xexpanded_weight(kw), tag="xexp")
    if bn is None:
        return C.conv2d_fwd(desc, xe, wp, conv.bias)[0]
    y, stats = C.conv2d_fwd(desc, xe, wp, None, want_stats=not bn_frozen(bn, training))
    scale, shift, _, _ = bn_finalize(stats, desc.N * desc.Hout * desc.Wout, bn, training, conv.bias)
    return bn_apply(y, scale, shift, act, out=y)


def head_stage(feat, convs, acts, mul=1.0):
    """ReflectionPad2d(3)+Conv7x7 heads with tiny Cout on a channels_last half feature map.

    ``convs`` is a list of nn.Conv2d reading the SAME feature map (e.g. flow and weight heads,
    networks.py:170-171); they are evaluated as one separable convolution: a 1x7 MFMA pass
    producing Cout*7 fp32 row responses, then a 7-tap vertical gather with bias and the output
    non-linearity.  ``acts``: per output channel 0 linear*mul, 1 tanh, 2 sigmoid.
    Returns NCHW fp32 [N, sum Cout, H, W]."""
    kh, kw = convs[0].kernel_size
    cout = sum(c.out_channels for c in convs)
    n, _, h, w = feat.shape
    desc = C.make_desc(tuple(feat.shape), (cout * kh + 7) // 8 * 8, (1, kw), 1, (0, kw // 2), C.PAD_REFLECT, feat.dtype,
                       out_f32=True)
    holder = convs[0]
    key = ("ysplit", desc.dtype, feat.shape[1]) + tuple((c.weight._version, c.weight.data_ptr()) for c in convs)
    cache = holder.__dict__.setdefault("_ir2rgb_packed", {})
    hit = cache.get("ysplit")
    if hit is None or hit[0] != key:
        with torch.no_grad():
            wcat = torch.cat([c.weight.detach().float() for c in convs], 0)
            if wcat.shape[1] != feat.shape[1]:      # the feature map runs at a padded width (autograd.padded_width)
                wcat = torch.cat([wcat, wcat.new_zeros((wcat.shape[0], feat.shape[1] - wcat.shape[1]) + tuple(wcat.shape[2:]))], 1)
            cache["ysplit"] = (key, C.pack_weight(desc, _ysplit_weight(wcat).contiguous()))
            if SC.ENABLED:
                SC.produced(cache["ysplit"][1], "packed head weight")
    wp = cache["ysplit"][1]
    if SC.ENABLED:
        SC.consumed(wp, "packed head weight")
    t, _ = C.conv2d_fwd(desc, feat, wp, None, want_stats=False)
    bias = torch.cat([c.bias.detach().float() for c in convs], 0).contiguous()
    out = torch.empty((n, cout, h, w), dtype=torch.float32, device=feat.device)
    packed_acts = 0
    for i, a in enumerate(acts):
        packed_acts |= (a & 15) << (4 * i)
    with _lib.on_device(feat):
        rc = _lib.lib().ir2rgb_head_finish(t, bias, out, n, h, w, cout, kh, desc.Cout, kh // 2,
                                           packed_acts, float(mul), _lib.current_stream(feat))
    _lib.check(rc, "head_finish")
    return out


def warp_blend(raw, prev, flow, weight, want_warp=False):
    """img_final = raw*w + grid_sample(prev[:, -3:], grid+flow)*(1-w)  (networks.py:207-209)."""
    for t in (raw, prev, flow, weight):
        _lib.require_device(t, dtype=torch.float32)
    n, _, h, w = raw.shape
    out = torch.empty_like(raw)
    warp = torch.empty_like(raw) if want_warp else None
    with _lib.on_device(raw):
        rc = _lib.lib().ir2rgb_warp_blend_fwd(raw, prev, flow, weight, out, warp, n,
                                              prev.shape[1], h, w, _lib.current_stream(raw))
    _lib.check(rc, "warp_blend_fwd")
    return (out, warp) if want_warp else out


_UNIT = {}


def bn_apply_add(a, b):
    """a + b on channels_last half tensors (bn_apply with unit scale / zero shift)."""
    ch = a.shape[1]
    key = (ch, a.device)
    if key not in _UNIT:
        _UNIT[key] = (torch.ones(ch, dtype=torch.float32, device=a.device),
                      torch.zeros(ch, dtype=torch.float32, device=a.device))
        # shared by every later call on ANY stream (the generators run independent branches on two): the fills above
        # were queued on the current stream only, so make them visible to all streams once
        torch.cuda.synchronize(a.device)
        if SC.ENABLED:
            SC.produced(_UNIT[key][0], "unit scale"), SC.produced(_UNIT[key][1], "zero shift")
    one, zero = _UNIT[key]
    if SC.ENABLED:
        SC.consumed(one, "unit scale"), SC.consumed(zero, "zero shift")
    return bn_apply(a, one, zero, ACT_NONE, res1=b)
