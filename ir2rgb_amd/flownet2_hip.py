"""FlowNet2's convolution stacks on the hand-written MFMA kernels (SURVEY section 8 "next" row f1).

Executes the parameter tree of ``ir2rgb_amd.flownet2_pytorch.models.FlowNet2`` (same module names
as reference models/flownet2_pytorch/networks/{FlowNetC,FlowNetS,FlowNetSD,FlowNetFusion}.py) with
``ir2rgb_conv2d_fwd``: every ``conv(...)+LeakyReLU(0.1)``, ``deconv`` (ConvTranspose 4x4 s2),
``i_conv`` and ``predict_flow`` becomes one launch with the activation fused, and every ``torch.cat``
of the refinement decoders disappears: producers write straight into channel slices of NHWC
concatenation buffers (ir2rgb_conv_desc.ldy / co_off) and consumers read slices (ldx / ci_off).
Concatenation widths that are not multiples of 64 (473, 1026, 770, 386, 194, 162, 82) are padded with
zero channels in the buffer and zero columns in the packed weights.

The 2-channel ConvTranspose2d(2,2,4,2,1) flow up-samplers are a small direct kernel writing into the
same buffers.  Left on torch (tiny 2..12-channel fp32 tensors): the x4 interpolations and the
normalisation / concatenation glue of FlowNet2.forward.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib
from . import conv as C
from . import layers as L

LEAKY01 = 2  # ir2rgb_conv_desc.act code for LeakyReLU(0.1)
_DT = {torch.bfloat16: 1, torch.float16: 2}


def _p(t):

    # (a plain int: accepted by the fastcall bindings and by ctypes' c_void_p parameters alike; a c_void_p object per
    # argument cost 0.75 us, ten of them per launch)
    return t.data_ptr() if t is not None else 0


def _up64(c):
    return (c + 63) // 64 * 64


class View:
    """Channels [off, off+ch) of an NHWC half buffer ``buf`` (logical [N, ld, H, W], channels_last)."""

    def __init__(self, buf, off, ch):
        self.buf, self.off, self.ch = buf, off, ch

    @property
    def n(self):
        return self.buf.shape[0]

    @property
    def ld(self):
        return self.buf.shape[1]

    @property
    def hw(self):
        return self.buf.shape[2], self.buf.shape[3]


class Workspace:
    """Per-(sub-network, input shape) pool of NHWC buffers, allocated and zero-filled once.  Buffers are
    handed out in call order; pad channels are never written, so they stay zero across calls, and every
    real channel is overwritten by its producer on each forward."""
    _current = None

    def __init__(self):
        self.bufs, self.cursor = [], 0

    @classmethod
    def activate(cls, net, key):
        """The pool of (net, key), created on first use.  Pools live ON the sub-network module (not in a
        table keyed by id(net): ids are recycled once a network is garbage collected)."""
        pools = net.__dict__.setdefault("_ir2rgb_workspaces", {})
        ws = pools.get(key)
        if ws is None:
            ws = pools[key] = Workspace()
        ws.cursor = 0
        cls._current = ws
        return ws

    def take(self, shape, dtype, device):
        if self.cursor < len(self.bufs):
            b = self.bufs[self.cursor]
            if tuple(b.shape) != tuple(shape) or b.dtype != dtype or b.device != device:
                raise RuntimeError("FlowNet2 workspace: allocation sequence changed")
        else:
            b = torch.zeros(shape, dtype=dtype, device=device).contiguous(memory_format=torch.channels_last)
            self.bufs.append(b)
        self.cursor += 1
        return b


def new_buf(n, ld, h, w, dtype, device):
    if Workspace._current is not None:
        return Workspace._current.take((n, ld, h, w), dtype, device)
    return torch.zeros((n, ld, h, w), dtype=dtype, device=device).contiguous(memory_format=torch.channels_last)


def _pad_cin(cin_to, transposed):
    def f(w):
        ci = w.shape[0] if transposed else w.shape[1]
        if ci == cin_to:
            return w
        if transposed:
            return torch.cat([w, w.new_zeros((cin_to - ci,) + tuple(w.shape[1:]))], 0)
        return torch.cat([w, w.new_zeros((w.shape[0], cin_to - ci) + tuple(w.shape[2:]))], 1)
    return f


def conv(xv, mod, yv, act, *, k=None, stride=None, pad=None, transposed=False, out_f32=False, weight_fn=None, tag="fn"):
    """yv <- act(conv(xv)).  xv / yv are Views; ``mod`` an nn.Conv2d / nn.ConvTranspose2d (weights, bias)."""
    k = tuple(mod.kernel_size) if k is None else k
    stride = tuple(mod.stride) if stride is None else stride
    pad = tuple(mod.padding) if pad is None else pad
    n = xv.n
    h, w = xv.hw
    cin = _up64(xv.ch)
    if xv.off + cin > xv.ld:
        raise ValueError("input view: padded channel range exceeds the buffer")
    dt = xv.buf.dtype
    cout = yv.ch
    oh, ow = yv.hw
    desc = C.ConvDesc(n, h, w, cin, oh, ow, cout, k[0], k[1], stride[0], stride[1], pad[0], pad[1], C.PAD_ZERO,
                      int(transposed), _DT[dt], act, int(out_f32), xv.ld, xv.off, yv.ld, yv.off)
    fn = _pad_cin(cin, transposed) if weight_fn is None else weight_fn
    wp = L.packed_weight(mod, desc, fn, tag=tag)
    C.conv2d_fwd_view(desc, xv.buf, wp, mod.bias, yv.buf)
    return yv


def first_conv(x_nchw, mod, yv, dtype):
    """Small-Cin first layer (7x7 s2 on 3/12 channels, 3x3 on 6/11): x-im2col + (kh x 1) MFMA convolution."""
    kh, kw = mod.kernel_size
    sh, sw = mod.stride
    ph, pw = mod.padding
    cx = 64 if mod.in_channels * kw <= 64 else 128
    xe = L.xexpand(x_nchw, kw, sw, pw, C.PAD_ZERO, dtype, cx=cx)
    return conv(View(xe, 0, cx), mod, yv, LEAKY01, k=(kh, 1), stride=(sh, 1), pad=(ph, 0),
                weight_fn=L._xexpanded_weight(kw, cx), tag="fn_xexp")


def put_nchw(x_nchw, yv, act=0):
    """NCHW fp32 -> channel slice of an NHWC half buffer (optionally LeakyReLU(0.1))."""
    x = x_nchw.float().contiguous()
    n, c, h, w = x.shape
    with _lib.on_device(x):
        rc = _lib.lib().ir2rgb_nchw_f32_to_nhwc_half_slice(x, yv.buf, n, c, h, w, yv.ld, yv.off, act,
                                                           _DT[yv.buf.dtype], _lib.current_stream(x))
    _lib.check(rc, "nchw_f32_to_nhwc_half_slice")


def corr_nhwc(mod, av, bv, yv, slope):
    """FlowNetC's cost volume (Correlation(20, 1, 20, 1, 2), FlowNetC.py:27) + LeakyReLU straight from the two NHWC half
    feature maps into a channel slice of ``yv.buf`` (ir2rgb_correlation_nhwc_half: banded MFMA products).  False when the
    operator's parameters or the shape are outside what that kernel covers."""
    cfg = tuple(getattr(mod, k, None) for k in ("pad_size", "kernel_size", "max_displacement", "stride1", "stride2"))
    if cfg != (20, 1, 20, 1, 2) or getattr(mod, "corr_multiply", 1) != 1 or yv.ch != 441 or av.ch != bv.ch:
        return False
    import os
    if os.environ.get("IR2RGB_CORR_MFMA", "1") == "0":
        return False
    n = av.n
    h, w = av.hw
    with _lib.on_device(av.buf):
        rc = _lib.lib().ir2rgb_correlation_nhwc_half(av.buf, av.ld, av.off, bv.buf, bv.ld, bv.off, yv.buf, 1, yv.ld,
                                                     yv.off, float(slope), n, av.ch, h, w, _DT[av.buf.dtype],
                                                     _lib.current_stream(av.buf))
    if rc == -2:        # IR2RGB_ENOSUP
        return False
    _lib.check(rc, "correlation_nhwc_half")
    return True


def flow_up(flow, mod, yv):
    """ConvTranspose2d(2,2,4,2,1) of a 2-channel fp32 flow, written into a 2-channel slice of ``yv.buf``."""
    n, _, h, w = flow.shape
    with _lib.on_device(flow):
        rc = _lib.lib().ir2rgb_flow_upsample_slice(flow, mod.weight, mod.bias, yv.buf, n, h, w, yv.ld, yv.off,
                                                   _DT[yv.buf.dtype], _lib.current_stream(flow))
    _lib.check(rc, "flow_upsample_slice")


def predict(xv, mod, n, h, w):
    """predict_flow (3x3, 2 channels): fp32 NCHW [N,2,h,w]."""
    out = torch.empty((n, 2, h, w), dtype=torch.float32, device=xv.buf.device).contiguous(memory_format=torch.channels_last)
    conv(xv, mod, View(out, 0, 2), 0, out_f32=True)
    return out.contiguous()


class _Decoder:
    """Coarse-to-fine refinement shared by FlowNetC / FlowNetS / FlowNetSD: levels 6 -> 2."""
    SKIP = {5: 512, 4: 512, 3: 256, 2: 128}
    DEC = {5: 512, 4: 256, 3: 128, 2: 64}

    def __init__(self, n, h6, w6, dtype, device):
        self.cat = {}
        for lvl in (5, 4, 3, 2):
            s = 2 ** (6 - lvl)
            width = self.SKIP[lvl] + self.DEC[lvl] + 2
            self.cat[lvl] = new_buf(n, _up64(width), h6 * s, w6 * s, dtype, device)

    def skip_view(self, lvl):
        return View(self.cat[lvl], 0, self.SKIP[lvl])

    def full_view(self, lvl):
        return View(self.cat[lvl], 0, self.SKIP[lvl] + self.DEC[lvl] + 2)

    def run(self, net, feat6, inter=False):
        n = feat6.n
        feat, flow_in = feat6, feat6
        for lvl in (6, 5, 4, 3):
            h, w = feat.hw
            flow = predict(flow_in, getattr(net, f"predict_flow{lvl}"), n, h, w)
            nxt = self.cat[lvl - 1]
            so, do = self.SKIP[lvl - 1], self.DEC[lvl - 1]
            conv(feat, getattr(net, f"deconv{lvl - 1}")[0], View(nxt, so, do), LEAKY01, transposed=True)
            flow_up(flow, getattr(net, f"upsampled_flow{lvl}_to_{lvl - 1}"), View(nxt, so + do, 2))
            feat = self.full_view(lvl - 1)
            if inter:
                ic = getattr(net, f"inter_conv{lvl - 1}")[0]
                ibuf = new_buf(n, _up64(ic.out_channels), nxt.shape[2], nxt.shape[3], nxt.dtype, nxt.device)
                flow_in = conv(feat, ic, View(ibuf, 0, ic.out_channels), 0)
            else:
                flow_in = feat
        h, w = feat.hw
        return predict(flow_in, net.predict_flow2, n, h, w)


def _dense(n, ch, h, w, dtype, device):
    return View(new_buf(n, _up64(ch), h, w, dtype, device), 0, ch)


def flownetc(net, x, dtype):
    """x [N,6,H,W] fp32 (two normalised images) -> flow2 [N,2,H/4,W/4] fp32 (reference FlowNetC.py:75-126)."""
    Workspace.activate(net, (tuple(x.shape), dtype, x.device))
    n, _, H, W = x.shape
    dev = x.device
    dec = _Decoder(n, H // 64, W // 64, dtype, dev)
    feats = []
    for img, skip in ((x[:, 0:3], True), (x[:, 3:6], False)):
        c1 = first_conv(img, net.conv1[0], _dense(n, 64, H // 2, W // 2, dtype, dev), dtype)
        c2 = conv(c1, net.conv2[0], dec.skip_view(2) if skip else _dense(n, 128, H // 4, W // 4, dtype, dev), LEAKY01)
        feats.append(conv(c2, net.conv3[0], _dense(n, 256, H // 8, W // 8, dtype, dev), LEAKY01))
    a3, b3 = feats
    merged = new_buf(n, 512, H // 8, W // 8, dtype, dev)                           # [redir 32 | corr 441 | pad]
    conv(a3, net.conv_redir[0], View(merged, 0, 32), LEAKY01)
    if not corr_nhwc(net.corr, a3, b3, View(merged, 32, 441), 0.1):                # MFMA cost volume + corr_activation
        cost = net.corr(L.to_nchw_f32(a3.buf), L.to_nchw_f32(b3.buf))             # (shapes it does not cover: scalar kernel, fp32)
        put_nchw(cost, View(merged, 32, 441), act=LEAKY01)
    c3 = conv(View(merged, 0, 473), net.conv3_1[0], dec.skip_view(3), LEAKY01)
    c4 = conv(conv(c3, net.conv4[0], _dense(n, 512, H // 16, W // 16, dtype, dev), LEAKY01), net.conv4_1[0], dec.skip_view(4), LEAKY01)
    c5 = conv(conv(c4, net.conv5[0], _dense(n, 512, H // 32, W // 32, dtype, dev), LEAKY01), net.conv5_1[0], dec.skip_view(5), LEAKY01)
    c6 = conv(conv(c5, net.conv6[0], _dense(n, 1024, H // 64, W // 64, dtype, dev), LEAKY01), net.conv6_1[0],
              _dense(n, 1024, H // 64, W // 64, dtype, dev), LEAKY01)
    return dec.run(net, c6)


def flownets(net, x, dtype):
    """x [N,12,H,W] fp32 -> flow2 (reference FlowNetS.py:57-93)."""
    Workspace.activate(net, (tuple(x.shape), dtype, x.device))
    n, _, H, W = x.shape
    dev = x.device
    dec = _Decoder(n, H // 64, W // 64, dtype, dev)
    c1 = first_conv(x, net.conv1[0], _dense(n, 64, H // 2, W // 2, dtype, dev), dtype)
    c2 = conv(c1, net.conv2[0], dec.skip_view(2), LEAKY01)
    c3 = conv(conv(c2, net.conv3[0], _dense(n, 256, H // 8, W // 8, dtype, dev), LEAKY01), net.conv3_1[0], dec.skip_view(3), LEAKY01)
    c4 = conv(conv(c3, net.conv4[0], _dense(n, 512, H // 16, W // 16, dtype, dev), LEAKY01), net.conv4_1[0], dec.skip_view(4), LEAKY01)
    c5 = conv(conv(c4, net.conv5[0], _dense(n, 512, H // 32, W // 32, dtype, dev), LEAKY01), net.conv5_1[0], dec.skip_view(5), LEAKY01)
    c6 = conv(conv(c5, net.conv6[0], _dense(n, 1024, H // 64, W // 64, dtype, dev), LEAKY01), net.conv6_1[0],
              _dense(n, 1024, H // 64, W // 64, dtype, dev), LEAKY01)
    return dec.run(net, c6)


def flownetsd(net, x, dtype):
    """x [N,6,H,W] fp32 -> flow2 (reference FlowNetSD.py:66-105)."""
    Workspace.activate(net, (tuple(x.shape), dtype, x.device))
    n, _, H, W = x.shape
    dev = x.device
    dec = _Decoder(n, H // 64, W // 64, dtype, dev)
    c0 = first_conv(x, net.conv0[0], _dense(n, 64, H, W, dtype, dev), dtype)
    c1 = conv(conv(c0, net.conv1[0], _dense(n, 64, H // 2, W // 2, dtype, dev), LEAKY01), net.conv1_1[0],
              _dense(n, 128, H // 2, W // 2, dtype, dev), LEAKY01)
    c2 = conv(conv(c1, net.conv2[0], _dense(n, 128, H // 4, W // 4, dtype, dev), LEAKY01), net.conv2_1[0], dec.skip_view(2), LEAKY01)
    c3 = conv(conv(c2, net.conv3[0], _dense(n, 256, H // 8, W // 8, dtype, dev), LEAKY01), net.conv3_1[0], dec.skip_view(3), LEAKY01)
    c4 = conv(conv(c3, net.conv4[0], _dense(n, 512, H // 16, W // 16, dtype, dev), LEAKY01), net.conv4_1[0], dec.skip_view(4), LEAKY01)
    c5 = conv(conv(c4, net.conv5[0], _dense(n, 512, H // 32, W // 32, dtype, dev), LEAKY01), net.conv5_1[0], dec.skip_view(5), LEAKY01)
    c6 = conv(conv(c5, net.conv6[0], _dense(n, 1024, H // 64, W // 64, dtype, dev), LEAKY01), net.conv6_1[0],
              _dense(n, 1024, H // 64, W // 64, dtype, dev), LEAKY01)
    return dec.run(net, c6, inter=True)


def flownetfusion(net, x, dtype):
    """x [N,11,H,W] fp32 -> flow0 [N,2,H,W] (reference FlowNetFusion.py:47-66)."""
    Workspace.activate(net, (tuple(x.shape), dtype, x.device))
    n, _, H, W = x.shape
    dev = x.device
    cat0 = new_buf(n, 128, H, W, dtype, dev)            # [conv0 64 | deconv0 16 | flow1_up 2 | pad]
    cat1 = new_buf(n, 192, H // 2, W // 2, dtype, dev)  # [conv1 128 | deconv1 32 | flow2_up 2 | pad]
    c0 = first_conv(x, net.conv0[0], View(cat0, 0, 64), dtype)
    c1 = conv(conv(c0, net.conv1[0], _dense(n, 64, H // 2, W // 2, dtype, dev), LEAKY01), net.conv1_1[0], View(cat1, 0, 128), LEAKY01)
    c2 = conv(conv(c1, net.conv2[0], _dense(n, 128, H // 4, W // 4, dtype, dev), LEAKY01), net.conv2_1[0],
              _dense(n, 128, H // 4, W // 4, dtype, dev), LEAKY01)
    flow2 = predict(c2, net.predict_flow2, n, H // 4, W // 4)
    flow_up(flow2, net.upsampled_flow2_to_1, View(cat1, 160, 2))
    conv(c2, net.deconv1[0], View(cat1, 128, 32), LEAKY01, transposed=True)
    i1 = conv(View(cat1, 0, 162), net.inter_conv1[0], _dense(n, 32, H // 2, W // 2, dtype, dev), 0)
    flow1 = predict(i1, net.predict_flow1, n, H // 2, W // 2)
    flow_up(flow1, net.upsampled_flow1_to_0, View(cat0, 80, 2))
    conv(View(cat1, 0, 162), net.deconv0[0], View(cat0, 64, 16), LEAKY01, transposed=True)
    i0 = conv(View(cat0, 0, 82), net.inter_conv0[0], _dense(n, 16, H, W, dtype, dev), 0)
    return predict(i0, net.predict_flow0, n, H, W)
