"""Guard-band test of the kernels' addressing (a memory access fault was on record for a
revision whose convolutions staged operands with flat ``global_load_lds`` and whose weight gradients went through MIOpen;
DESIGN.md section 11 has the analysis).  Today's kernels stage through buffer resources with exact extents -- an
out-of-range lane reads zeros and cannot fault -- and everything else is plain loads behind explicit bounds.  This test
checks both halves of that claim without provoking a fault:

* every INPUT lives inside a larger allocation whose neighbourhood (64 KB before and after) is NaN: a kernel that reads
  outside its operand and lets the value reach a result produces NaN / a different result;
* every OUTPUT lives inside a larger allocation whose neighbourhood holds a canary pattern: a kernel that writes outside
  its result changes a canary.

Results must be bit-identical to the same call on ordinary tensors.  Shapes are chosen so that tiles overhang every
edge (pixel counts and channel counts that are not multiples of the tile sizes, last workgroup mostly empty).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

GUARD = 32768     # elements on each side


class Guarded:
    """A tensor of ``shape`` (channels_last if 4-D and ``nhwc``) inside a guard-banded allocation."""

    def __init__(self, like, fill):
        t = like.detach()
        self.nhwc = t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()
        n = t.numel()
        self.buf = torch.full((n + 2 * GUARD,), fill, dtype=t.dtype, device=t.device)
        core = self.buf[GUARD:GUARD + n]
        if self.nhwc:
            N, C, H, W = t.shape
            self.t = core.view(N, H, W, C).permute(0, 3, 1, 2)
        else:
            self.t = core.view(t.shape)
        self.t.copy_(t)
        self.fill = fill

    def intact(self):
        lo, hi = self.buf[:GUARD], self.buf[-GUARD:]
        if self.fill != self.fill:      # NaN
            return bool(torch.isnan(lo).all() and torch.isnan(hi).all())
        return bool((lo == self.fill).all() and (hi == self.fill).all())


def _nhwc(n, c, h, w, dtype, dev, gen):
    return torch.randn(n, c, h, w, generator=gen).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)


CONV_CASES = {
    # name: (N, Cin, H, W, Cout, k, stride, pad, pad_mode, transposed, output_padding)
    "patch3x3_reflect_overhang": (1, 256, 37, 67, 256, 3, 1, 1, 1, False, 0),
    "igemm3x3_zero_small_cin": (2, 64, 19, 23, 192, 3, 1, 1, 0, False, 0),
    "d4x4_s2_p2": (1, 64, 33, 65, 128, 4, 2, 2, 0, False, 0),
    "down3x3_s2": (1, 128, 21, 35, 256, 3, 2, 1, 0, False, 0),
    "up3x3_transposed": (1, 256, 9, 13, 128, 3, 2, 1, 0, True, 1),
    "head1x7_reflect": (1, 128, 11, 29, 24, (1, 7), 1, (0, 3), 1, False, 0),
}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name", list(CONV_CASES))
def test_convolution_forward_and_weight_gradient_stay_in_bounds(dev, name, dtype):
    from ir2rgb_amd import conv as C
    n, cin, h, w, cout, k, stride, pad, pad_mode, transposed, op = CONV_CASES[name]
    gen = torch.Generator().manual_seed(len(name))
    x = _nhwc(n, cin, h, w, dtype, dev, gen)
    kk = C._pair(k)
    wshape = (cin, cout) + kk if transposed else (cout, cin) + kk
    wt = (torch.randn(wshape, generator=gen) * 0.05).to(dev)
    bias = torch.randn(cout, generator=gen).to(dev)
    desc = C.make_desc(tuple(x.shape), cout, k, stride, pad, pad_mode, dtype, transposed, op, out_f32=(cout % 8 != 0 or name.startswith("head")))
    wp = C.pack_weight(desc, wt)
    y0, s0 = C.conv2d_fwd(desc, x, wp, bias, want_stats=True)
    gx, gw, gb = Guarded(x, float("nan")), Guarded(wp, float("nan")), Guarded(bias, float("nan"))
    gy = Guarded(y0, 7.0)
    gy.t.zero_()
    y1, s1 = C.conv2d_fwd(desc, gx.t, gw.t, gb.t, want_stats=True, out=gy.t)
    torch.cuda.synchronize()
    assert torch.isfinite(y1.float()).all() and torch.equal(y1, y0) and torch.equal(s1, s0)
    assert gy.intact(), "convolution wrote outside its output"
    assert gx.intact() and gw.intact() and gb.intact()
    if name.startswith("head"):
        return
    g = _nhwc(n, cout, y0.shape[2], y0.shape[3], dtype, dev, gen)
    dw0 = C.conv2d_wgrad(desc, x, g)
    gg = Guarded(g, float("nan"))
    dw1 = C.conv2d_wgrad(desc, gx.t, gg.t)
    torch.cuda.synchronize()
    assert torch.isfinite(dw1).all() and torch.equal(dw1, dw0)


def _fwd_ws_guarded(desc, x, wp, y, stats):
    """ir2rgb_conv2d_fwd_ws called directly with a guard-banded split-K workspace of exactly
    ir2rgb_conv2d_fwd_workspace_bytes: zeros inside (the tickets start even), a canary pattern on both sides."""
    import ctypes
    from ir2rgb_amd import _lib
    from ir2rgb_amd import conv as C
    lib = _lib.lib()
    n = int(lib.ir2rgb_conv2d_fwd_workspace_bytes(ctypes.byref(desc)))
    assert n >= 0
    band = 1 << 16
    buf = torch.full((n + 2 * band,), 0x5A, dtype=torch.uint8, device=x.device)
    buf[band:band + n].zero_()
    ws = buf[band:band + n]
    rc = lib.ir2rgb_conv2d_fwd_ws(ctypes.byref(desc), C._p(x), C._p(wp), C._p(None), C._p(y), C._p(stats),
                                  C._p(ws) if n else C._p(None), n, _lib.current_stream(x))
    _lib.check(rc, "conv2d_fwd_ws")
    torch.cuda.synchronize()
    return n, bool((buf[:band] == 0x5A).all() and (buf[band + n:] == 0x5A).all())


# name: (channels, H, W, expects a split-K workspace) -- shapes that MUST select the patch-staged kernel's in-place
# reflect adjoint: H even, W a multiple of its 64-pixel tile, >= 200 tiles.  The first is the dominant kernel of the
# training bench (roofline.kernel: 1024 -> 1024 @32x64, split variant 4), the second its unsplit form.
ADJ_CASES = {"split_1024ch_32x64": (1024, 32, 64, True), "unsplit_256ch_64x128": (256, 64, 128, False)}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name", list(ADJ_CASES))
def test_reflect_adjoint_stays_in_bounds(dev, name, dtype):
    """The in-place reflect adjoint of conv3x3_patch_kernel (and, in the split form, its partial-tile hand-over through the
    caller's workspace): operands inside NaN bands, output and workspace inside canary bands, bit-identical results."""
    from ir2rgb_amd import conv as C
    ch, h, w, split = ADJ_CASES[name]
    gen = torch.Generator().manual_seed(3)
    gyv = _nhwc(1, ch, h, w, dtype, dev, gen)
    wt = (torch.randn(ch, ch, 3, 3, generator=gen) * 0.03).to(dev)
    dadj = C.make_desc(gyv.shape, ch, 3, 1, 1, C.PAD_REFLECT_ADJ, dtype)
    assert C.kernel_name(dadj) == "conv3x3_patch_kernel"
    wp = C.pack_weight(C.make_desc(gyv.shape, ch, 3, 1, 1, C.PAD_ZERO, dtype), wt, adjoint=True)
    d0, _ = C.conv2d_fwd(dadj, gyv, wp)
    gi, gw = Guarded(gyv, float("nan")), Guarded(wp, float("nan"))
    go = Guarded(d0, 7.0)
    go.t.zero_()
    nbytes, ws_intact = _fwd_ws_guarded(dadj, gi.t, gw.t, go.t, None)
    assert (nbytes > 0) == split, "split-K form expected for the 1024-channel shape only"
    assert ws_intact, "the split-K hand-over wrote outside its workspace"
    assert torch.isfinite(go.t.float()).all() and torch.equal(go.t, d0) and go.intact()
    assert gi.intact() and gw.intact()
    # the forward twin of the same shape (reflect padding, BatchNorm partial sums) through the same guarded path
    dfw = C.make_desc(gyv.shape, ch, 3, 1, 1, C.PAD_REFLECT, dtype)
    assert C.kernel_name(dfw) == "conv3x3_patch_kernel"
    wf = C.pack_weight(dfw, wt)
    y0, s0 = C.conv2d_fwd(dfw, gyv, wf, None, want_stats=True)
    gwf, gy, gs = Guarded(wf, float("nan")), Guarded(y0, 7.0), Guarded(s0, 7.0)
    gy.t.zero_()
    _, ws_intact = _fwd_ws_guarded(dfw, gi.t, gwf.t, gy.t, gs.t)
    assert ws_intact and torch.equal(gy.t, y0) and torch.equal(gs.t, s0) and gy.intact() and gs.intact()


LINE_CASES = {
    # name: (N, Cin, H, W, Cout, (kh, kw), (sh, sw), (ph, pw), reflect)  -- the layers of the round-3 kernels
    "first7x1_reflect_ragged": (2, 64, 21, 130, 64, (7, 1), (1, 1), (3, 0), 1),     # conv7x1_col forward, conv_wgrad_col 7x1
    "first7x1_two_cout_tiles": (1, 64, 13, 96, 128, (7, 1), (1, 1), (3, 0), 1),
    "head1x7_wgrad": (2, 64, 9, 200, 64, (1, 7), (1, 1), (0, 3), 1),                # conv_wgrad_line 1x7
    "d4x1_stride2": (3, 64, 37, 65, 64, (4, 1), (2, 1), (2, 0), 0),                 # conv_wgrad_col 4x1 stride 2
}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name", list(LINE_CASES))
def test_line_kernels_stay_in_bounds(dev, name, dtype):
    """The k x 1 / 1 x k kernels of round 3 (column-tile forward, line / column weight gradients with up to 256 split
    slabs and the wide finish pass): operands inside NaN guard bands, the weight-gradient workspace of EXACTLY
    ir2rgb_conv2d_wgrad_workspace_elems and the destination inside canary bands, results bit-identical to plain tensors."""
    from ir2rgb_amd import _lib
    from ir2rgb_amd import conv as C
    n, cin, h, w, cout, k, stride, pad, reflect = LINE_CASES[name]
    gen = torch.Generator().manual_seed(len(name) + 3)
    x = _nhwc(n, cin, h, w, dtype, dev, gen)
    desc = C.make_desc(tuple(x.shape), cout, k, stride, pad, C.PAD_REFLECT if reflect else C.PAD_ZERO, dtype)
    wt = (torch.randn((cout, cin) + k, generator=gen) * 0.05).to(dev)
    wp = C.pack_weight(desc, wt)
    gx, gw = Guarded(x, float("nan")), Guarded(wp, float("nan"))
    y0, s0 = C.conv2d_fwd(desc, x, wp, None, want_stats=True)
    gy = Guarded(y0, 7.0)
    gy.t.zero_()
    y1, s1 = C.conv2d_fwd(desc, gx.t, gw.t, None, want_stats=True, out=gy.t)
    torch.cuda.synchronize()
    assert torch.isfinite(y1.float()).all() and torch.equal(y1, y0) and torch.equal(s1, s0)
    assert gy.intact() and gx.intact() and gw.intact()
    # weight gradient through the C ABI with a guarded workspace of exactly the declared size and a guarded destination
    g = _nhwc(n, cout, y0.shape[2], y0.shape[3], dtype, dev, gen)
    gg = Guarded(g, float("nan"))
    dw0 = C.conv2d_wgrad(desc, x, g)
    lib = _lib.lib()
    nws = lib.ir2rgb_conv2d_wgrad_workspace_elems(desc)
    assert nws > 0
    ws = Guarded(torch.zeros(nws, device=dev), 5.0)
    dst = Guarded(torch.zeros_like(dw0), 3.0)
    rc = lib.ir2rgb_conv2d_wgrad(desc, gx.t, gg.t, dst.t, ws.t, _lib.current_stream(x))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(dst.t, dw0) and torch.isfinite(dw0).all()
    assert ws.intact(), "weight-gradient kernel wrote outside its workspace"
    assert dst.intact(), "finish pass wrote outside the weight gradient"
    assert gx.intact() and gg.intact()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_batchnorm_kernels_stay_in_bounds(dev, dtype):
    from ir2rgb_amd import autograd as A
    from ir2rgb_amd import layers as L
    gen = torch.Generator().manual_seed(5)
    y = _nhwc(1, 128, 37, 53, dtype, dev, gen)
    r1 = _nhwc(1, 128, 37, 53, dtype, dev, gen)
    scale, shift = torch.rand(128, generator=gen).to(dev) + 0.5, torch.randn(128, generator=gen).to(dev)
    mean, invstd = torch.randn(128, generator=gen).to(dev) * 0.1, torch.rand(128, generator=gen).to(dev) + 0.5
    z0 = L.bn_apply(y, scale, shift, L.ACT_RELU, r1)
    gyv, gr, gs, gh = Guarded(y, float("nan")), Guarded(r1, float("nan")), Guarded(scale, float("nan")), Guarded(shift, float("nan"))
    gz = Guarded(z0, 7.0)
    z1 = L.bn_apply(gyv.t, gs.t, gh.t, L.ACT_RELU, gr.t, out=gz.t)
    torch.cuda.synchronize()
    assert torch.equal(z1, z0) and gz.intact()
    g = _nhwc(1, 128, 37, 53, dtype, dev, gen)
    a0 = A.bn_bwd(g, y, scale, shift, mean, invstd, L.ACT_RELU)
    gg, gm, gi = Guarded(g, float("nan")), Guarded(mean, float("nan")), Guarded(invstd, float("nan"))
    a1 = A.bn_bwd(gg.t, gyv.t, gs.t, gh.t, gm.t, gi.t, L.ACT_RELU)
    torch.cuda.synchronize()
    for u, v in zip(a0, a1):
        assert torch.isfinite(v.float()).all() and torch.equal(u, v)


def test_operators_stay_in_bounds(dev):
    """The three FlowNet2 operators and the warp-blend on guard-banded fp32 operands (flows that leave the image)."""
    from ir2rgb_amd import ext
    from ir2rgb_amd import layers as L
    gen = torch.Generator().manual_seed(9)
    img, flow = torch.rand(2, 3, 37, 53, generator=gen).to(dev), (torch.randn(2, 2, 37, 53, generator=gen) * 30).to(dev)
    out0 = torch.zeros_like(img)
    ext.resample2d_cuda.forward(img, flow, out0, 1)
    gi, gf = Guarded(img, float("nan")), Guarded(flow, float("nan"))
    go = Guarded(out0, 7.0)
    go.t.zero_()
    ext.resample2d_cuda.forward(gi.t, gf.t, go.t, 1)
    torch.cuda.synchronize()
    assert torch.equal(go.t, out0) and go.intact() and torch.isfinite(out0).all()
    raw, prev, wgt = torch.rand(2, 3, 37, 53, generator=gen).to(dev), torch.rand(2, 6, 37, 53, generator=gen).to(dev), \
        torch.rand(2, 1, 37, 53, generator=gen).to(dev)
    b0 = L.warp_blend(raw, prev, flow, wgt)
    b1 = L.warp_blend(Guarded(raw, float("nan")).t, Guarded(prev, float("nan")).t, gf.t, Guarded(wgt, float("nan")).t)
    torch.cuda.synchronize()
    assert torch.equal(b0, b1) and torch.isfinite(b1).all()
    f1, f2 = torch.randn(1, 32, 9, 13, generator=gen).to(dev), torch.randn(1, 32, 9, 13, generator=gen).to(dev)
    from ir2rgb_amd.flownet2_pytorch.networks.correlation_package.correlation import Correlation
    corr = Correlation(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1)
    c0 = corr(f1, f2)
    c1 = corr(Guarded(f1, float("nan")).t, Guarded(f2, float("nan")).t)
    torch.cuda.synchronize()
    assert torch.equal(c0, c1) and torch.isfinite(c1).all()
