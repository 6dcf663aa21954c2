"""Python face of the MFMA convolution entry points of libir2rgb_hip.so.

Activations are NHWC half precision.  On the torch side an NHWC buffer is represented as a
logical NCHW tensor in ``torch.channels_last`` memory format (same bytes), so tensors can be
handed to/from ordinary torch code without copies.
"""
import ctypes

import torch

from . import _lib
from . import streamcheck as SC
from ._lib import ConvDesc

BF16, F16 = 1, 2
_TORCH2DT = {torch.bfloat16: BF16, torch.float16: F16}
PAD_ZERO, PAD_REFLECT = 0, 1
PAD_REFLECT_ADJ = 2   # the launch is the data gradient of a reflection-padded 3x3 conv (see include/ir2rgb_hip.h)
ACT_NONE, ACT_LEAKY02 = 0, 1


def _p(t):

    # (a plain int: accepted by the fastcall bindings and by ctypes' c_void_p parameters alike; a c_void_p object per
    # argument cost 0.75 us, ten of them per launch)
    return t.data_ptr() if t is not None else 0


def out_size(h, k, stride, pad, transposed=False, output_padding=0):
    if transposed:
        return (h - 1) * stride - 2 * pad + k + output_padding
    return (h + 2 * pad - k) // stride + 1


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def make_desc(x_shape, cout, k, stride, pad, pad_mode, dtype, transposed=False, output_padding=0, act=ACT_NONE,
              out_f32=False, ldx=0, ci_off=0, ldy=0, co_off=0, stats_per_sample=False):
    """x_shape = (N, Cin, H, W) logical; k / stride / pad = int or (h, w) pair."""
    # Descriptors are immutable by convention and a layer asks for the same one every window: one object per distinct
    # argument tuple, which also carries what is derived from it once (``_addr`` for the fastcall bindings, workspace
    # sizes, statistics rows, kernel name) -- building the struct and asking the library again cost ~5 us per launch.
    key = (x_shape, cout, k, stride, pad, pad_mode, dtype, transposed, output_padding, act, out_f32, ldx, ci_off, ldy, co_off,
           stats_per_sample)
    try:
        hit = _DESCS.get(key)
    except TypeError:           # (an unhashable shape, e.g. a list)
        key, hit = None, None
    if hit is not None:
        return hit
    n, cin, h, w = x_shape
    (kh, kw), (sh, sw), (ph, pw) = _pair(k), _pair(stride), _pair(pad)
    ho = out_size(h, kh, sh, ph, transposed, output_padding)
    wo = out_size(w, kw, sw, pw, transposed, output_padding)
    d = sealed(ConvDesc(n, h, w, cin, ho, wo, cout, kh, kw, sh, sw, ph, pw, pad_mode, int(transposed), _TORCH2DT[dtype],
                        act, int(out_f32), ldx, ci_off, ldy, co_off, int(stats_per_sample)))
    if key is not None:
        if len(_DESCS) > 4096:
            _DESCS.clear()
        _DESCS[key] = d
    return d


_DESCS = {}


def sealed(desc):
    """A finished descriptor: ``_addr`` = its address (what the fastcall bindings of libir2rgb_hip.so take)."""
    desc._addr = ctypes.addressof(desc)
    return desc


_KERNEL_NAMES = {}


def kernel_name(desc):
    """Device kernel ir2rgb_conv2d_fwd would launch for ``desc`` ('' if it cannot run it); cached per descriptor."""
    name = getattr(desc, "_kname", None)
    if name is None:
        key = bytes(desc)
        name = _KERNEL_NAMES.get(key)
        if name is None:
            name = _KERNEL_NAMES[key] = _lib.lib().ir2rgb_conv2d_kernel_name(ctypes.byref(desc)).decode()
        desc._kname = name
    return name


def is_nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def empty_nhwc(n, c, h, w, dtype, device):
    return torch.empty((n, c, h, w), dtype=dtype, device=device, memory_format=torch.channels_last)


class PackBatch:
    """Every (descriptor, fp32 weight, packed buffer, adjoint) job of a network refreshed by one launch
    (``run``), in place.  The device table is built once; it holds raw pointers, so the weight and
    packed tensors are kept alive here and must not be reallocated."""

    def __init__(self, jobs):
        lib = _lib.lib()
        self.keep = [(w, p) for _, w, p, _ in jobs]
        arr = (_lib.PackJob * len(jobs))()
        for j, (desc, w, packed, adjoint) in enumerate(jobs):
            _lib.require_device(w, dtype=torch.float32)
            _lib.require_device(packed)
            arr[j].desc, arr[j].w, arr[j].wpacked, arr[j].adjoint = desc, w.data_ptr(), packed.data_ptr(), int(bool(adjoint))
        self.dtype = jobs[0][0].dtype
        self.device = jobs[0][1].device
        nbytes = lib.ir2rgb_conv2d_pack_batch_table_bytes(arr, len(jobs))
        if nbytes < 0:
            _lib.check(int(nbytes), "conv2d_pack_batch_table_bytes")
        host = torch.zeros(int(nbytes), dtype=torch.uint8)
        nblocks = ctypes.c_int(0)
        n = lib.ir2rgb_conv2d_pack_batch_build(arr, len(jobs), host.data_ptr(), int(nbytes),
                                               ctypes.byref(nblocks))
        if n < 0:
            _lib.check(int(n), "conv2d_pack_batch_build")
        self.nentries, self.nblocks = int(n), int(nblocks.value)
        self.table = host.to(self.device)

    def run(self):
        with _lib.on_device(self.table):
            rc = _lib.lib().ir2rgb_conv2d_pack_batch_run(self.table, self.nentries, self.nblocks, self.dtype,
                                                         _lib.current_stream(self.table))
        _lib.check(rc, "conv2d_pack_batch_run")


def pack_weight(desc, weight, adjoint=False):
    """weight: fp32 torch layout ([Cout,Cin,kh,kw], or [Cin,Cout,kh,kw] for transposed).  adjoint=True:
    ``desc`` is the data-gradient convolution of a stride-1 Conv2d and ``weight`` its forward weight."""
    _lib.require_device(weight, dtype=torch.float32)
    lib = _lib.lib()
    n = lib.ir2rgb_conv2d_packed_weight_elems(desc)
    if n < 0:
        _lib.check(int(n), "conv2d_packed_weight_elems")
    dt = torch.bfloat16 if desc.dtype == BF16 else torch.float16
    packed = torch.empty(n, dtype=dt, device=weight.device)
    with _lib.on_device(weight):
        fn = lib.ir2rgb_conv2d_pack_weight_adjoint if adjoint else lib.ir2rgb_conv2d_pack_weight
        rc = fn(desc, weight, packed, _lib.current_stream(weight))
    _lib.check(rc, "conv2d_pack_weight")
    return packed


def stats_rows(desc):
    r = getattr(desc, "_stats_rows", None)
    if r is None:
        r = _lib.lib().ir2rgb_conv2d_stats_rows(desc)
        if r < 0:
            _lib.check(r, "conv2d_stats_rows")
        desc._stats_rows = r
    return r


# Optional per-launch timing (bench.py's roofline leg): when PROFILE is a dict, convolutions are
# bracketed by HIP events on the stream they are launched on and tallied by shape -- every shape, or
# only PROFILE["only"] when that key is set (an event pair costs a queue marker, so the timed region
# of bench.py brackets just the dominant shape).  Launches issued while weight-gradient work is in
# flight on the side stream (SIDE_BUSY, set by ir2rgb_amd.autograd) share the GPU and are skipped: the
# events would time the overlap, not the kernel.
PROFILE = None
SIDE_BUSY = False


def _prof_key(desc):
    return (desc.Cin, desc.Hin, desc.Win, desc.Cout, desc.kh, desc.kw, desc.stride_h, desc.pad_mode, desc.transposed)


def _prof_begin(desc, tag=None):
    prof = PROFILE
    if prof is None or SIDE_BUSY or torch.cuda.is_current_stream_capturing():
        return None   # (events recorded while a HIP graph is being captured are graph nodes, not timers)
    key = _prof_key(desc) if tag is None else _prof_key(desc) + (tag,)
    only = prof.get("only")
    if only is not None and only != key:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return prof, key, e0


def _prof_end(tok, desc, name=None):
    prof, key, e0 = tok
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    rec = prof.setdefault("shapes", {}).get(key)
    if rec is None:
        name = name or _lib.lib().ir2rgb_conv2d_kernel_name(ctypes.byref(desc)).decode()
        rec = prof["shapes"][key] = {"flops": _flops(desc), "events": [], "kernel": name}
    rec["events"].append((e0, e1))


def _flops(desc):
    if desc.transposed:
        taps = desc.kh * desc.kw / float(desc.stride_h * desc.stride_w)
    else:
        taps = desc.kh * desc.kw
    return 2.0 * desc.N * desc.Hout * desc.Wout * desc.Cout * desc.Cin * taps


# Workspace of the split-K form of the 3x3 kernel (ir2rgb_conv2d_fwd_ws): one per (device, stream), zeroed once --
# launches on one stream are ordered, which is what sharing it needs.  While a HIP graph is being captured a fresh
# zeroed one is taken per launch instead (it belongs to the graph's memory pool; kernels captured on one stream may
# replay next to kernels of ANOTHER graph captured on the same stream).
_SPLIT_WS = {}


def _fwd_workspace(desc, x):
    n = getattr(desc, "_ws_bytes", None)         # (descriptors are built once per layer and shape and never edited)
    if n is None:
        n = desc._ws_bytes = int(_lib.lib().ir2rgb_conv2d_fwd_workspace_bytes(desc))
        if n < 0:
            _lib.check(n, "conv2d_fwd_workspace_bytes")
    if n == 0:
        return None, 0
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(n, dtype=torch.uint8, device=x.device), n
    k = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _SPLIT_WS.get(k)
    if ws is None or ws.numel() < n:
        ws = _SPLIT_WS[k] = torch.zeros(n, dtype=torch.uint8, device=x.device)
        if SC.ENABLED:
            SC.produced(ws, "split-K workspace")
    if SC.ENABLED:
        SC.consumed(ws, "split-K workspace")
    return ws, n


def conv2d_fwd(desc, x, wpacked, bias=None, want_stats=False, out=None):
    """x: logical [N,Cin,H,W] channels_last half tensor.  Returns (y, stats_partial | None)."""
    if not is_nhwc(x) or x.dtype not in _TORCH2DT or _TORCH2DT[x.dtype] != desc.dtype:
        raise ValueError("conv2d_fwd: x must be a channels_last half tensor of the descriptor's dtype")
    if tuple(x.shape) != (desc.N, desc.Cin, desc.Hin, desc.Win):
        raise ValueError(f"conv2d_fwd: x shape {tuple(x.shape)} does not match the descriptor")
    if not x.is_cuda:
        raise ValueError("conv2d_fwd: GPU tensors only (no CPU fallback)")
    odt = torch.float32 if desc.out_f32 else x.dtype
    y = out if out is not None else empty_nhwc(desc.N, desc.Cout, desc.Hout, desc.Wout, odt, x.device)
    stats = None
    if want_stats:
        stats = torch.empty((stats_rows(desc), 2, desc.Cout), dtype=torch.float32, device=x.device)
    with _lib.on_device(x):
        ws, ws_bytes = _fwd_workspace(desc, x)
        tok = _prof_begin(desc)
        rc = _lib.lib().ir2rgb_conv2d_fwd_ws(desc, x, wpacked, bias, y, stats, ws, ws_bytes,
                                             _lib.current_stream(x))
    _lib.check(rc, "conv2d_fwd")
    if tok is not None:
        _prof_end(tok, desc)
    return y, stats


def conv2d_wgrad(desc, x, gy, out=None, accumulate=False):
    """Weight gradient (fp32, torch weight layout) of the convolution ``desc`` from its forward input ``x`` and the
    gradient ``gy`` w.r.t. its output (both channels_last half).  ``out``: a contiguous fp32 tensor of the weight's shape
    to write into (e.g. a parameter's slice of a flat all-reduce buffer) instead of a fresh allocation.
    ``accumulate`` (with ``out``): out += the gradient, summed in the kernel's finish pass (ir2rgb_conv2d_wgrad_acc)."""
    if not (is_nhwc(x) and is_nhwc(gy)) or x.dtype != gy.dtype or _TORCH2DT.get(x.dtype) != desc.dtype:
        raise ValueError("conv2d_wgrad: x and gy must be channels_last half tensors of the descriptor's dtype")
    if tuple(x.shape) != (desc.N, desc.Cin, desc.Hin, desc.Win) or tuple(gy.shape) != (desc.N, desc.Cout, desc.Hout, desc.Wout):
        raise ValueError("conv2d_wgrad: shapes do not match the descriptor")
    lib = _lib.lib()
    if accumulate and out is None:
        raise ValueError("conv2d_wgrad: accumulate needs the tensor to add to (out)")
    n = getattr(desc, "_wgrad_ws_acc" if accumulate else "_wgrad_ws", None)
    if n is None:
        n = (lib.ir2rgb_conv2d_wgrad_acc_workspace_elems if accumulate else lib.ir2rgb_conv2d_wgrad_workspace_elems)(desc)
        if n < 0:
            _lib.check(int(n), "conv2d_wgrad_workspace_elems")
        setattr(desc, "_wgrad_ws_acc" if accumulate else "_wgrad_ws", n)
    ws = torch.empty(n, dtype=torch.float32, device=x.device)
    shape = (desc.Cin, desc.Cout, desc.kh, desc.kw) if desc.transposed else (desc.Cout, desc.Cin, desc.kh, desc.kw)
    if out is not None:
        if tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
            raise ValueError("conv2d_wgrad: out must be a contiguous fp32 tensor of the weight's shape on the inputs' device")
        dw = out
    else:
        dw = torch.empty(shape, dtype=torch.float32, device=x.device)
    with _lib.on_device(x):
        fn = lib.ir2rgb_conv2d_wgrad_acc if accumulate else lib.ir2rgb_conv2d_wgrad
        tok = _prof_begin(desc, "wgrad") if PROFILE is not None else None
        rc = fn(desc, x, gy, dw, ws, _lib.current_stream(x))
        if tok is not None:
            _prof_end(tok, desc, "conv_wgrad")
    _lib.check(rc, "conv2d_wgrad")
    if accumulate:
        return None
    # (out: a NEW tensor object over the same memory -- autograd adopts a gradient without cloning it only when nobody
    # else holds the object it is handed)
    return dw if out is None else out.view(shape)


def conv2d_fwd_view(desc, xbuf, wpacked, bias, ybuf, stats=None):
    """Raw launch on caller-owned buffers (channel-slice views: desc.ldx/ci_off/ldy/co_off).  No shape
    checks beyond device / alignment: the caller (flownet2_hip) owns the buffer geometry."""
    if not (xbuf.is_cuda and ybuf.is_cuda):
        raise ValueError("conv2d_fwd_view: GPU tensors only (no CPU fallback)")
    tok = _prof_begin(desc)
    with _lib.on_device(xbuf):
        rc = _lib.lib().ir2rgb_conv2d_fwd(desc, xbuf, wpacked, bias, ybuf, stats,
                                          _lib.current_stream(xbuf))
    _lib.check(rc, "conv2d_fwd")
    if tok is not None:
        _prof_end(tok, desc)
