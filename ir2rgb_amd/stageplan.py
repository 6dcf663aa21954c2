"""The common convolution stage with its host work done once: ``ConvStageFn``'s straight-line case as a cached plan.

``autograd.ConvStageFn`` handles every stage of the networks (first layers over x-expanded images, padded widths, thin fp32
outputs, sample groups, evaluation-mode BatchNorm, backward flags ...) and pays for that generality in Python: ~30 us of
interpreter time per forward and ~40 us per backward on top of the launches, 226 times per training window -- a third of the
host time that bounds the loop once the GPU needs less than the host.  Nine stages in ten are
the same plain case -- conv -> training-mode BatchNorm -> activation (+ residuals) on a 64-multiple power-of-two width, as one
sample group or as the discriminators' batch of groups (GroupedStagePlan, which also honours the backward flags) -- and
everything about such a stage except its tensors is a function of (module, input shape, spec):
descriptors (forward, data gradient, weight gradient), output shapes, statistics rows, workspace sizes, the split of the
BatchNorm launches.  A ``StagePlan`` computes those once and then issues the SAME library calls with the SAME arguments as
the general path, in a handful of statements (``IR2RGB_LEAN_STAGE=0`` switches it off; tests/test_stage_backward_gpu.py
holds the two paths against each other bit for bit, forward and backward).

Anything else -- and any call while the stream check or a side-stream weight gradient is on -- takes the general path.
"""
import os

import torch

from . import _lib
from . import conv as C
from . import layers as L
from . import streamcheck as SC

ENABLED = os.environ.get("IR2RGB_LEAN_STAGE", "1") != "0"
_F32 = torch.float32
_CL = torch.channels_last


class StagePlan:
    __slots__ = ("desc", "out_shape", "stats_shape", "rows", "cout", "cin", "count", "npix", "fused", "act", "dtc", "tdtype",
                 "ws_bytes", "nblk", "x_shape", "dgrad", "wdesc", "wshape", "wgrad_ws", "spec", "groups", "order", "dgrads")

    def __init__(self, x, spec, conv, bn, fused_bn):
        A = _autograd()
        dt = spec["dtype"]
        cout, cin = conv.out_channels, conv.in_channels
        self.spec = spec
        self.cout, self.cin, self.tdtype, self.dtc = cout, cin, dt, A._DT[dt]
        self.x_shape = tuple(x.shape)
        G = self.groups = spec.get("groups", 1)
        self.order = tuple(spec.get("group_order") or range(G))
        self.dgrads = {}
        d = self.desc = C.make_desc(self.x_shape, cout, spec["k"], spec["stride"], spec["pad"], spec["pad_mode"], dt,
                                    spec["transposed"], spec.get("output_padding", 0), act=0, out_f32=False,
                                    stats_per_sample=G > 1)
        if d.N % G:
            raise ValueError("conv stage: sample groups need N % groups == 0 and no residual inputs")
        self.out_shape = (d.N, d.Cout, d.Hout, d.Wout)
        self.rows = C.stats_rows(d)
        self.stats_shape = (self.rows, 2, cout)
        self.count = self.npix = d.N * d.Hout * d.Wout // G          # per sample group
        self.fused = bool(fused_bn and cout % 64 == 0 and self.rows // G <= L.FUSED_BN_MAX_ROWS)
        self.act = spec["act"]
        n = getattr(d, "_ws_bytes", None)
        if n is None:
            n = d._ws_bytes = int(_lib.lib().ir2rgb_conv2d_fwd_workspace_bytes(d))
            if n < 0:
                _lib.check(n, "conv2d_fwd_workspace_bytes")
        self.ws_bytes = n
        nblk = _lib.lib().ir2rgb_bn_bwd_blocks(self.npix, cout)
        if nblk < 0:
            _lib.check(nblk, "bn_bwd_blocks")
        self.nblk = nblk
        self.dgrad = None          # built at the first backward: (launch desc, pack desc, tag, adjoint, ws_bytes) | False
        self.wdesc = self.wshape = self.wgrad_ws = None

    # ------------------------------------------------------------------------------------------------------------
    def forward(self, ctx, x, bias, res1, res2, conv, bn):
        if x.dtype is not self.tdtype or not x.is_contiguous(memory_format=_CL):
            raise ValueError("conv stage: x must be a channels_last half tensor of the stage's dtype")
        lib, d, dev, cout = _lib.lib(), self.desc, x.device, self.cout
        stream = _lib.current_stream(x)
        wp = L.packed_weight(conv, d, None, "w")
        y = torch.empty(self.out_shape, dtype=self.tdtype, device=dev, memory_format=_CL)
        stats = torch.empty(self.stats_shape, dtype=_F32, device=dev)
        ws, wsb = C._fwd_workspace(d, x) if self.ws_bytes else (None, 0)
        tok = C._prof_begin(d) if C.PROFILE is not None else None
        rc = lib.ir2rgb_conv2d_fwd_ws(d, x, wp, None, y, stats, ws, wsb, stream)
        if rc:
            _lib.check(rc, "conv2d_fwd")
        if tok is not None:
            C._prof_end(tok, d)
        # BatchNorm: [scale | shift | mean | invstd] in one allocation, addressed by offset (no views)
        vec = torch.empty((4, cout), dtype=_F32, device=dev)
        pv = vec.data_ptr()
        z = torch.empty(self.out_shape, dtype=self.tdtype, device=dev, memory_format=_CL)
        _, pw, pb, prm, prv, has_rm, momentum, eps, trs = L._bn_ptrs(bn)
        track = trs and has_rm
        reps = L._STAT_UPDATES
        if self.fused:
            rc = lib.ir2rgb_bn_finalize_apply(stats, self.rows, cout, self.count, pw, pb, bias, prm if track else None,
                                              prv if track else None, momentum, eps, pv, pv + 4 * cout, pv + 8 * cout,
                                              pv + 12 * cout, reps, y, res1, res2, z, self.npix, self.act, self.dtc, stream)
            if rc:
                _lib.check(rc, "bn_finalize_apply")
        else:
            rc = lib.ir2rgb_bn_finalize_ex(stats, self.rows, cout, self.count, pw, pb, bias, prm if track else None,
                                           prv if track else None, momentum, eps, pv, pv + 4 * cout, pv + 8 * cout,
                                           pv + 12 * cout, reps, 0, stream)
            if rc:
                _lib.check(rc, "bn_finalize")
            rc = lib.ir2rgb_bn_apply(y, pv, pv + 4 * cout, res1, res2, z, self.npix, cout, self.act, self.dtc, stream)
            if rc:
                _lib.check(rc, "bn_apply")
        if track and bn.num_batches_tracked is not None:
            L._PENDING_COUNTERS.append((bn.num_batches_tracked, reps))
        ctx.plan = self
        ctx.conv = conv
        ctx.has_res = (res1 is not None, res2 is not None)
        ctx.save_for_backward(x, y, vec)
        return z

    # ------------------------------------------------------------------------------------------------------------
    def _build_dgrad(self, conv, n=None):
        """The data-gradient launch of conv_dgrad() for this stage (``n``: over the leading n samples only), or False where
        that function does more than one launch (reflection padding other than the in-place 3x3 adjoint: zero-padded
        convolution + fold pass)."""
        A = _autograd()
        spec, dt = self.spec, self.tdtype
        kh, kw = spec["k"]
        (sh, sw), (ph, pw) = spec["stride"], spec["pad"]
        _, cin, hin, win = self.x_shape
        n = self.x_shape[0] if n is None else n
        gshape = (n,) + tuple(self.out_shape[1:])
        if spec["transposed"]:
            desc = C.make_desc(gshape, cin, (kh, kw), (sh, sw), (ph, pw), C.PAD_ZERO, dt)
            plan = (desc, desc, "dgrad", False)
        elif sh == 1 and sw == 1:
            if spec["pad_mode"] == C.PAD_REFLECT:
                dadj = C.make_desc(gshape, cin, 3, 1, 1, C.PAD_REFLECT_ADJ, dt) if (kh, kw, ph, pw) == (3, 3, 1, 1) else None
                if dadj is None or C.kernel_name(dadj) != "conv3x3_patch_kernel":
                    return False
                plan = (dadj, C.make_desc(gshape, cin, 3, 1, 1, C.PAD_ZERO, dt), "dgrad", True)
            else:
                desc = C.make_desc(gshape, cin, (kh, kw), 1, (kh - 1 - ph, kw - 1 - pw), C.PAD_ZERO, dt)
                plan = (desc, desc, "dgrad", True)
        else:
            if spec["pad_mode"] != C.PAD_ZERO:
                return False
            hfull, wfull = (gshape[2] - 1) * sh - 2 * ph + kh, (gshape[3] - 1) * sw - 2 * pw + kw
            if not (0 <= hin - hfull < sh and 0 <= win - wfull < sw):
                return False
            key = (n, gshape[2], gshape[3], gshape[1], hin, win, cin, kh, kw, sh, sw, ph, pw, C.PAD_ZERO, 1, A._DT[dt], 0, 0, 0, 0, 0, 0)
            desc = A._ADJ_DESCS.get(key)
            if desc is None:
                desc = A._ADJ_DESCS[key] = C.sealed(C.ConvDesc(*key))
            plan = (desc, desc, "dgrad", False)
        d = plan[0]
        nb = getattr(d, "_ws_bytes", None)
        if nb is None:
            nb = d._ws_bytes = int(_lib.lib().ir2rgb_conv2d_fwd_workspace_bytes(d))
            if nb < 0:
                _lib.check(nb, "conv2d_fwd_workspace_bytes")
        return plan + (nb,)

    def _build_wgrad(self, conv):
        spec = self.spec
        self.wdesc = d = C.make_desc(self.x_shape, self.cout, spec["k"], spec["stride"], spec["pad"], spec["pad_mode"], self.tdtype,
                                     bool(spec["transposed"]), spec.get("output_padding", 0))
        self.wshape = (d.Cin, d.Cout, d.kh, d.kw) if d.transposed else (d.Cout, d.Cin, d.kh, d.kw)
        lib = _lib.lib()
        n, na = lib.ir2rgb_conv2d_wgrad_workspace_elems(d), lib.ir2rgb_conv2d_wgrad_acc_workspace_elems(d)
        if n < 0 or na < 0:
            _lib.check(int(min(n, na)), "conv2d_wgrad_workspace_elems")
        self.wgrad_ws = (n, na)

    def backward(self, ctx, gz):
        A = _autograd()
        x, y, vec = ctx.saved_tensors
        conv = ctx.conv
        lib, dev, cout, dt = _lib.lib(), y.device, self.cout, self.tdtype
        stream = _lib.current_stream(y)
        if gz.dtype is not dt or not gz.is_contiguous(memory_format=_CL):
            gz = A._as_half_nhwc(gz, dt)
        # ---- BatchNorm + activation backward: [dgamma | dbeta | partial rows] in one allocation
        buf = torch.empty((self.nblk * 2 + 5) * cout, dtype=_F32, device=dev)
        gy = torch.empty(self.out_shape, dtype=dt, device=dev, memory_format=_CL)
        pv, pbuf = vec.data_ptr(), buf.data_ptr()
        rc = lib.ir2rgb_bn_bwd(gz, y, pv, pv + 4 * cout, pv + 8 * cout, pv + 12 * cout, gy, pbuf, pbuf + 4 * cout,
                               pbuf + 8 * cout, self.npix, cout, self.act, self.dtc, stream)
        if rc:
            _lib.check(rc, "bn_bwd")
        need = ctx.needs_input_grad
        # ---- data gradient
        dx = None
        if need[0]:
            dg = self.dgrad
            if dg is None:
                dg = self.dgrad = self._build_dgrad(conv)
            if dg is False:
                dx = A.conv_dgrad(gy, conv, self.spec, self.x_shape, None)
            else:
                dd, dpack, tag, adjoint, nb = dg
                wp = L.packed_weight(conv, dpack, None, tag, adjoint)
                dx = torch.empty(self.x_shape, dtype=dt, device=dev, memory_format=_CL)
                ws, wsb = C._fwd_workspace(dd, gy) if nb else (None, 0)
                tok = C._prof_begin(dd) if C.PROFILE is not None else None
                rc = lib.ir2rgb_conv2d_fwd_ws(dd, gy, wp, None, dx, None, ws, wsb, stream)
                if rc:
                    _lib.check(rc, "conv2d_fwd")
                if tok is not None:
                    C._prof_end(tok, dd)
        # ---- weight gradient (same destinations as the general path: sink slice / in-kernel accumulation / fresh tensor)
        dw = None
        if need[1]:
            if self.wdesc is None:
                self._build_wgrad(conv)
            wd, w = self.wdesc, conv.weight
            sink = A.GRAD_SINKS.get(w) if A.GRAD_SINKS else None
            have = w.grad
            acc = (A.ACCUMULATE_IN_KERNEL and sink is None and have is not None and have.dtype is _F32 and have.is_contiguous()
                   and have.shape == w.shape)
            wsn = torch.empty(self.wgrad_ws[1 if acc else 0], dtype=_F32, device=dev)
            if acc:
                out = have
            elif sink is not None:
                if tuple(sink.shape) != self.wshape or sink.dtype is not _F32 or not sink.is_contiguous():
                    raise ValueError("conv2d_wgrad: out must be a contiguous fp32 tensor of the weight's shape on the inputs' device")
                out = sink
            else:
                out = torch.empty(self.wshape, dtype=_F32, device=dev)
            tok = C._prof_begin(wd, "wgrad") if C.PROFILE is not None else None
            rc = (lib.ir2rgb_conv2d_wgrad_acc if acc else lib.ir2rgb_conv2d_wgrad)(wd, x, gy, out, wsn, stream)
            if rc:
                _lib.check(rc, "conv2d_wgrad")
            if tok is not None:
                C._prof_end(tok, wd, "conv_wgrad")
            if not acc:
                dw = out if sink is None else sink.view(self.wshape)
        # (training-mode BatchNorm removes the per-channel mean: the bias gradient is exactly zero = None)
        dgamma, dbeta = buf[:cout], buf[cout:2 * cout]
        hr = ctx.has_res
        return dx, dw, None, dgamma, dbeta, (gz if hr[0] else None), (gz if hr[1] else None), None, None, None


class GroupedStagePlan(StagePlan):
    """The same for a batch of G independent forwards (the discriminators' real | generated | raw frames along N): one
    convolution, BatchNorm per sample group by offset, and in backward the two things only these stages meet -- flags
    (autograd.backward_flags: no parameter gradients in the generator's pass) and passes in which only the leading k groups
    carry a gradient (the work then runs on that leading part of the batch; the rest of the gradient tensor stays
    unwritten, nobody reads it)."""
    __slots__ = ()

    def forward(self, ctx, x, bias, res1, res2, conv, bn):
        if res1 is not None or res2 is not None:
            raise ValueError("conv stage: sample groups need N % groups == 0 and no residual inputs")
        if x.dtype is not self.tdtype or not x.is_contiguous(memory_format=_CL):
            raise ValueError("conv stage: x must be a channels_last half tensor of the stage's dtype")
        lib, d, dev, cout, G = _lib.lib(), self.desc, x.device, self.cout, self.groups
        stream = _lib.current_stream(x)
        wp = L.packed_weight(conv, d, None, "w")
        y = torch.empty(self.out_shape, dtype=self.tdtype, device=dev, memory_format=_CL)
        stats = torch.empty(self.stats_shape, dtype=_F32, device=dev)
        ws, wsb = C._fwd_workspace(d, x) if self.ws_bytes else (None, 0)
        tok = C._prof_begin(d) if C.PROFILE is not None else None
        rc = lib.ir2rgb_conv2d_fwd_ws(d, x, wp, None, y, stats, ws, wsb, stream)
        if rc:
            _lib.check(rc, "conv2d_fwd")
        if tok is not None:
            C._prof_end(tok, d)
        vec = torch.empty((4, G, cout), dtype=_F32, device=dev)      # [scale | shift | mean | invstd][group]
        z = torch.empty(self.out_shape, dtype=self.tdtype, device=dev, memory_format=_CL)
        _, pw, pb, prm, prv, has_rm, momentum, eps, trs = L._bn_ptrs(bn)
        track = trs and has_rm
        reps = L._STAT_UPDATES
        py, pz, ps, pv = y.data_ptr(), z.data_ptr(), stats.data_ptr(), vec.data_ptr()
        npix, rg, c4 = self.npix, self.rows // G, cout * 4
        per_y, per_s, gc4 = npix * cout * y.element_size(), rg * 2 * c4, G * c4
        for g in self.order:
            r = reps[g] if isinstance(reps, tuple) else reps
            sc = pv + g * c4
            if self.fused:
                rc = lib.ir2rgb_bn_finalize_apply(ps + g * per_s, rg, cout, npix, pw, pb, bias, prm if track else None,
                                                  prv if track else None, momentum, eps, sc, sc + gc4, sc + 2 * gc4, sc + 3 * gc4,
                                                  r, py + g * per_y, None, None, pz + g * per_y, npix, self.act, self.dtc, stream)
                if rc:
                    _lib.check(rc, "bn_finalize_apply")
            else:
                rc = lib.ir2rgb_bn_finalize_ex(ps + g * per_s, rg, cout, npix, pw, pb, bias, prm if track else None,
                                               prv if track else None, momentum, eps, sc, sc + gc4, sc + 2 * gc4, sc + 3 * gc4,
                                               r, 0, stream)
                if rc:
                    _lib.check(rc, "bn_finalize")
                rc = lib.ir2rgb_bn_apply(py + g * per_y, sc, sc + gc4, None, None, pz + g * per_y, npix, cout, self.act, self.dtc,
                                         stream)
                if rc:
                    _lib.check(rc, "bn_apply")
            if track and bn.num_batches_tracked is not None:
                L._PENDING_COUNTERS.append((bn.num_batches_tracked, r))
        ctx.plan = self
        ctx.conv = conv
        ctx.has_res = (False, False)
        ctx.save_for_backward(x, y, vec)
        return z

    def backward(self, ctx, gz):
        A = _autograd()
        x, y, vec = ctx.saved_tensors
        conv = ctx.conv
        flags = getattr(conv, "_ir2rgb_bwd", 0)
        want_params = not (flags & A.SKIP_PARAM_GRADS)
        G, k = self.groups, getattr(conv, "_ir2rgb_active", None)
        n_full = self.x_shape[0]
        if k is not None and k < G:
            if want_params:
                raise RuntimeError("conv stage: a pass with inactive sample groups cannot produce parameter gradients")
            ka = k
        else:
            ka = G
        na = n_full // G * ka
        lib, dev, cout, dt = _lib.lib(), y.device, self.cout, self.tdtype
        stream = _lib.current_stream(y)
        if gz.dtype is not dt or not gz.is_contiguous(memory_format=_CL):
            gz = A._as_half_nhwc(gz, dt)
        # ---- BatchNorm + activation backward of the active groups (one partial-row region: same stream, one after the other)
        buf = torch.empty((self.nblk * 2 + 5) * cout, dtype=_F32, device=dev)
        gshape = (na,) + tuple(self.out_shape[1:])
        gy = torch.empty(gshape, dtype=dt, device=dev, memory_format=_CL)
        pv, pb_, c4 = vec.data_ptr(), buf.data_ptr(), cout * 4
        gc4, npix = G * c4, self.npix
        per = npix * cout * y.element_size()
        pg, py_, pgy = gz.data_ptr(), y.data_ptr(), gy.data_ptr()
        for g in range(ka):
            sc = pv + g * c4
            rc = lib.ir2rgb_bn_bwd(pg + g * per, py_ + g * per, sc, sc + gc4, sc + 2 * gc4, sc + 3 * gc4, pgy + g * per, pb_,
                                   pb_ + c4, pb_ + 2 * c4, npix, cout, self.act | (32 if g else 0), self.dtc, stream)
            if rc:
                _lib.check(rc, "bn_bwd")
        need = ctx.needs_input_grad
        # ---- data gradient over the active samples, into a tensor of the full batch's shape
        dx = None
        if need[0]:
            dg = self.dgrads.get(na)
            if dg is None:
                dg = self.dgrads[na] = self._build_dgrad(conv, na)
            xs = (na,) + tuple(self.x_shape[1:])
            if dg is False:
                dx = torch.empty(self.x_shape, dtype=dt, device=dev, memory_format=_CL)
                A.conv_dgrad(gy, conv, self.spec, xs, None, out=dx[:na])
            else:
                dd, dpack, tag, adjoint, nb = dg
                wp = L.packed_weight(conv, dpack, None, tag, adjoint)
                dx = torch.empty(self.x_shape, dtype=dt, device=dev, memory_format=_CL)
                ws, wsb = C._fwd_workspace(dd, gy) if nb else (None, 0)
                tok = C._prof_begin(dd) if C.PROFILE is not None else None
                rc = lib.ir2rgb_conv2d_fwd_ws(dd, gy, wp, None, dx, None, ws, wsb, stream)
                if rc:
                    _lib.check(rc, "conv2d_fwd")
                if tok is not None:
                    C._prof_end(tok, dd)
        # ---- weight gradient (whole batch: a pass that wants parameter gradients has every group active)
        dw = None
        if need[1] and want_params:
            if self.wdesc is None:
                self._build_wgrad(conv)
            wd, w = self.wdesc, conv.weight
            sink = A.GRAD_SINKS.get(w) if A.GRAD_SINKS else None
            have = w.grad
            acc = (A.ACCUMULATE_IN_KERNEL and sink is None and have is not None and have.dtype is _F32 and have.is_contiguous()
                   and have.shape == w.shape)
            wsn = torch.empty(self.wgrad_ws[1 if acc else 0], dtype=_F32, device=dev)
            if acc:
                out = have
            elif sink is not None:
                if tuple(sink.shape) != self.wshape or sink.dtype is not _F32 or not sink.is_contiguous():
                    raise ValueError("conv2d_wgrad: out must be a contiguous fp32 tensor of the weight's shape on the inputs' device")
                out = sink
            else:
                out = torch.empty(self.wshape, dtype=_F32, device=dev)
            tok = C._prof_begin(wd, "wgrad") if C.PROFILE is not None else None
            rc = (lib.ir2rgb_conv2d_wgrad_acc if acc else lib.ir2rgb_conv2d_wgrad)(wd, x, gy, out, wsn, stream)
            if rc:
                _lib.check(rc, "conv2d_wgrad")
            if tok is not None:
                C._prof_end(tok, wd, "conv_wgrad")
            if not acc:
                dw = out if sink is None else sink.view(self.wshape)
        if want_params:
            dgamma, dbeta = buf[:cout], buf[cout:2 * cout]
        else:
            dgamma = dbeta = None
        return dx, dw, None, dgamma, dbeta, None, None, None, None, None


def _autograd():
    from . import autograd
    return autograd


def lookup(x, spec, conv, bn, fused_bn):
    """The plan of this stage call, or None when the general path has to run it."""
    if not ENABLED or SC.ENABLED or bn is None or spec["first"] or not spec["training"]:
        return None
    A = _autograd()
    if A.WGRAD_SIDE_STREAM:
        return None
    plans = conv.__dict__.get("_ir2rgb_plans")
    if plans is None:
        plans = conv.__dict__["_ir2rgb_plans"] = {}
    key = (x.shape, spec["act"], spec["pad_mode"], spec["dtype"], spec["stride"], spec["pad"], spec["transposed"],
           spec["output_padding"], spec["out_f32"], spec["fused_leaky"], spec["fused_relu"], fused_bn, id(bn), spec["groups"],
           spec["group_order"])
    plan = plans.get(key)
    if plan is None:
        cout, cin = conv.out_channels, conv.in_channels
        ok = (not spec["out_f32"] and not spec["fused_leaky"] and not spec["fused_relu"] and A.padded_width(cout) == cout
              and A.padded_width(cin) == cin and x.dim() == 4 and x.shape[1] == cin and x.is_cuda
              and isinstance(bn, torch.nn.Module))
        if ok and spec["groups"] > 1:
            ok = x.shape[0] % spec["groups"] == 0
        plan = plans[key] = (GroupedStagePlan if spec["groups"] > 1 else StagePlan)(x, spec, conv, bn, fused_bn) if ok else False
        if len(plans) > 64:
            plans.clear()
    return plan or None
