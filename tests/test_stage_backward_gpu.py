"""GPU unit tests of each fused stage's backward (HIP BN/activation backward, HIP data gradient via
the adjoint convolution, fold/x-im2col adjoints, weight gradient) against torch fp32 autograd of the
same stage evaluated on the same half-rounded inputs and weights.

One stage deep, so only rounding separates the two: relative L2 <= 1e-2 (f16), 4e-2 (bf16)
(measured 2e-3..8e-3 and 1e-2..3e-2; LeakyReLU/ReLU masks flip for a few elements under rounding).
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()


def _leaf(t):
    return t.detach().clone().requires_grad_()


CASES = {
    # name: (Cin, H, W, Cout, k, stride, pad, pad_mode, transposed, opad, act, use_bn, with_res)
    "res3x3_reflect_bn_relu_res": (64, 12, 20, 64, 3, 1, 1, 1, False, 0, 1, True, True),
    "down3x3_s2_bn_relu": (64, 16, 24, 128, 3, 2, 1, 0, False, 0, 1, True, False),
    "up3x3T_s2_bn_relu": (128, 8, 12, 64, 3, 2, 1, 0, True, 1, 1, True, False),
    "d4x4_s2_p2_bn_leaky_odd": (64, 17, 21, 128, 4, 2, 2, 0, False, 0, 2, True, False),
    "d4x4_s2_p2_bn_leaky_even": (64, 16, 20, 128, 4, 2, 2, 0, False, 0, 2, True, False),
    "d4x4_s1_p2_bn_leaky": (128, 9, 13, 256, 4, 1, 2, 0, False, 0, 2, True, False),
}


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1.5e-2), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("name", list(CASES))
def test_stage_backward(dev, name, dtype, tol):
    from ir2rgb_amd import autograd as A
    from ir2rgb_amd import conv as C
    cin, h, w, cout, k, s, p, pm, tr, op, act, use_bn, with_res = CASES[name]
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    conv = (nn.ConvTranspose2d(cin, cout, k, s, p, output_padding=op) if tr else nn.Conv2d(cin, cout, k, s, 0 if pm else p)).to(dev)
    bn = nn.BatchNorm2d(cout).to(dev) if use_bn else None
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (1.0 / np.sqrt(cin * k * k)))
        conv.weight.copy_(conv.weight.to(dtype).float())  # exactly representable in half
        conv.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        if bn is not None:
            bn.weight.copy_(1 + 0.2 * torch.randn(cout, generator=g))
            bn.bias.copy_(0.2 * torch.randn(cout, generator=g))
    x0 = torch.randn(2, cin, h, w, generator=g).to(dev).to(dtype)
    xh = x0.contiguous(memory_format=torch.channels_last).requires_grad_()
    # reference (fp32 autograd)
    xr = _leaf(x0.float())
    wr, br = _leaf(conv.weight), _leaf(conv.bias)
    if tr:
        y = F.conv_transpose2d(xr, wr, br, s, p, op)
    elif pm:
        y = F.conv2d(F.pad(xr, (p,) * 4, mode="reflect"), wr, br, s)
    else:
        y = F.conv2d(xr, wr, br, s, p)
    gr, btr = _leaf(bn.weight), _leaf(bn.bias)
    z = F.batch_norm(y, None, None, gr, btr, True, 0.1, 1e-5)
    z = F.relu(z) if act == 1 else F.leaky_relu(z, 0.2)
    res0 = torch.randn(z.shape, generator=g).to(dev).to(dtype) if with_res else None
    rr = _leaf(res0.float()) if with_res else None
    if with_res:
        z = z + rr
    proj = torch.randn(z.shape, generator=g).to(dev)
    (z * proj).sum().backward()
    # HIP
    rh = res0.contiguous(memory_format=torch.channels_last).requires_grad_() if with_res else None
    zh = A.conv_stage(xh, conv, bn, act, pm, dtype, pad=p if pm else None, transposed=tr, output_padding=op, res1=rh)
    assert _rel(zh, z) <= tol
    (zh.float() * proj).sum().backward()
    errs = {"dx": _rel(xh.grad, xr.grad), "dW": _rel(conv.weight.grad, wr.grad), "dgamma": _rel(bn.weight.grad, gr.grad),
            "dbeta": _rel(bn.bias.grad, btr.grad)}
    if with_res:
        errs["dres"] = _rel(rh.grad, rr.grad)
    assert conv.bias.grad is None or conv.bias.grad.abs().max().item() == 0  # exactly 0 behind BatchNorm
    bad = {k2: v for k2, v in errs.items() if not v <= tol}
    assert not bad, f"{name}: {errs}"


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-2), (torch.bfloat16, 4e-2)])
def test_first_and_last_discriminator_layers_backward(dev, dtype, tol):
    """x-im2col first layer (Conv4x4 s2 p2 + LeakyReLU on a 6-channel NCHW fp32 image, input gradient
    needed: the generator loss flows through it) and the 1-channel fp32 logit layer."""
    from ir2rgb_amd import autograd as A
    from ir2rgb_amd import conv as C
    g = torch.Generator().manual_seed(9)
    c0 = nn.Conv2d(6, 64, 4, 2, 2).to(dev)
    c1 = nn.Conv2d(64, 1, 4, 1, 2).to(dev)
    with torch.no_grad():
        for c in (c0, c1):
            c.weight.copy_((torch.randn(c.weight.shape, generator=g) * 0.1).to(dtype).float())
    x0 = torch.randn(2, 6, 19, 26, generator=g).to(dev).to(dtype).float()
    xr = _leaf(x0)
    w0, b0, w1, b1 = _leaf(c0.weight), _leaf(c0.bias), _leaf(c1.weight), _leaf(c1.bias)
    hr = F.leaky_relu(F.conv2d(xr, w0, b0, 2, 2), 0.2)
    outr = F.conv2d(hr, w1, b1, 1, 2)
    proj = torch.randn(outr.shape, generator=g).to(dev)
    ((outr * proj).sum() + hr.sum() * 0.01).backward()
    xh = _leaf(x0)
    hh = A.conv_stage(xh, c0, None, 0, C.PAD_ZERO, dtype, first=True, fused_leaky=True)
    outh = A.conv_stage(hh, c1, None, 0, C.PAD_ZERO, dtype, out_f32=True)
    assert outh.dtype == torch.float32 and _rel(outh, outr) <= tol
    ((outh * proj).sum() + hh.float().sum() * 0.01).backward()
    errs = {"dx": _rel(xh.grad, xr.grad), "dW0": _rel(c0.weight.grad, w0.grad), "db0": _rel(c0.bias.grad, b0.grad),
            "dW1": _rel(c1.weight.grad, w1.grad), "db1": _rel(c1.bias.grad, b1.grad)}
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, errs


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 1, 35, 67), (3, 1, 67, 131), (2, 3, 5, 9), (1, 8, 4, 4)])
def test_thin_grad_expand(dev, dtype, shape):
    """ir2rgb_thin_grad_expand against the torch ops it replaced (cast, zero-padded channel copies, sum):
    the half copies are bit-exact, the sum agrees to fp32 summation order."""
    from ir2rgb_amd import autograd as A
    torch.manual_seed(5)
    gz = torch.randn(shape, device=dev)
    g64, g8, dbias = A.thin_grad_expand(gz, dtype)
    n, c, h, w = shape
    for got, ch in ((g64, 64), (g8, 8)):
        assert got.shape == (n, ch, h, w) and got.is_contiguous(memory_format=torch.channels_last)
        want = torch.zeros((n, ch, h, w), dtype=dtype, device=dev)
        want[:, :c] = gz.to(dtype)
        assert torch.equal(got, want)
    ref = gz.double().sum((0, 2, 3))
    assert (dbias.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    with pytest.raises(ValueError):
        A.thin_grad_expand(torch.randn(1, 9, 4, 4, device=dev), dtype)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-2), (torch.bfloat16, 4e-2)])
def test_generator_first_layer_backward(dev, dtype, tol):
    """ReflectionPad2d(3)+Conv7x7+BN+ReLU on a 9-channel image: weight / BN gradients (no input grad)."""
    from ir2rgb_amd import autograd as A
    from ir2rgb_amd import conv as C
    g = torch.Generator().manual_seed(10)
    conv, bn = nn.Conv2d(9, 64, 7).to(dev), nn.BatchNorm2d(64).to(dev)
    with torch.no_grad():
        conv.weight.copy_((torch.randn(conv.weight.shape, generator=g) * 0.05).to(dtype).float())
    x0 = torch.randn(1, 9, 20, 28, generator=g).to(dev).to(dtype).float()
    wr, gr, btr = _leaf(conv.weight), _leaf(bn.weight), _leaf(bn.bias)
    zr = F.relu(F.batch_norm(F.conv2d(F.pad(x0, (3,) * 4, mode="reflect"), wr, conv.bias), None, None, gr, btr, True))
    proj = torch.randn(zr.shape, generator=g).to(dev)
    (zr * proj).sum().backward()
    zh = A.conv_stage(x0, conv, bn, 1, C.PAD_REFLECT, dtype, first=True, pad=3)
    assert _rel(zh, zr) <= tol
    (zh.float() * proj).sum().backward()
    errs = {"dW": _rel(conv.weight.grad, wr.grad), "dgamma": _rel(bn.weight.grad, gr.grad), "dbeta": _rel(bn.bias.grad, btr.grad)}
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, errs


def test_fold_reflect_and_xexpand_adjoints(dev):
    """<A x, y> == <x, A^T y> for the two linear maps whose adjoints are hand-written."""
    from ir2rgb_amd import autograd as A
    from ir2rgb_amd import layers as L
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 64, 9, 11, generator=g).to(dev)
    y = torch.randn(2, 64, 13, 15, generator=g).to(dev)
    lhs = (F.pad(x, (2, 2, 2, 2), mode="reflect") * y).sum()
    yt = A.fold_reflect(y.to(torch.float16).contiguous(memory_format=torch.channels_last), 2).float()
    rhs = (x * yt).sum()
    assert abs(lhs.item() - rhs.item()) <= 5e-3 * abs(lhs.item()) + 0.5
    for (kw, s, p, pm) in [(7, 1, 3, 1), (4, 2, 2, 0)]:
        img = torch.randn(2, 6, 10, 21, generator=g).to(dev)
        e = L.xexpand(img, kw, s, p, pm, torch.float16)
        d = torch.randn(e.shape, generator=g).to(dev).to(torch.float16).contiguous(memory_format=torch.channels_last)
        d[:, 6 * kw:] = 0
        lhs = (e.float() * d.float()).sum()
        rhs = (img.half().float() * A.xexpand_bwd(d, 6, 21, kw, s, p, pm)).sum()
        assert abs(lhs.item() - rhs.item()) <= 5e-3 * abs(lhs.item()) + 0.5, (kw, lhs.item(), rhs.item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # Cin, H, W, Cout, k, stride, pad, pad_mode name, transposed, opad, act, residual
    (256, 16, 64, 256, 3, 1, 1, "PAD_REFLECT", False, 0, "ACT_RELU", True),      # residual block (in-place reflect adjoint, fused BN)
    (128, 32, 64, 128, 3, 1, 1, "PAD_REFLECT", False, 0, "ACT_NONE", True),
    (64, 64, 128, 128, 3, 2, 1, "PAD_ZERO", False, 0, "ACT_RELU", False),        # down-sampling (strided adjoint)
    (128, 16, 32, 64, 3, 2, 1, "PAD_ZERO", True, 1, "ACT_RELU", False),          # up-sampling (sub-pixel classes)
    (64, 70, 200, 128, 4, 2, 2, "PAD_ZERO", False, 0, "ACT_LEAKY", False),       # discriminator layer, > 128 statistics rows
    (64, 20, 24, 64, 7, 1, 3, "PAD_REFLECT", False, 0, "ACT_RELU", False),       # reflect 7x7: data gradient via the fold pass
])
def test_planned_stage_equals_the_general_stage(dev, dtype, case):
    """ir2rgb_amd/stageplan.py (the plain stage with its host work cached) against ConvStageFn's general code: the same
    library calls on the same arguments, so outputs, BatchNorm buffers and every gradient agree bit for bit -- two forwards
    and backwards each (the second one hits the cached plan; the weight gradient accumulates in the kernel)."""
    import copy
    from ir2rgb_amd import autograd as A, conv as C, layers as L, stageplan
    cin, h, w, cout, k, s, p, pm, tr, op, act, use_res = case
    gen = torch.Generator().manual_seed(cin + h + k)
    x0 = torch.randn(2, cin, h, w, generator=gen).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    conv0 = (torch.nn.ConvTranspose2d(cin, cout, k, s, p, output_padding=op) if tr else torch.nn.Conv2d(cin, cout, k, s, 0)).to(dev)
    bn0 = torch.nn.BatchNorm2d(cout).to(dev)
    with torch.no_grad():
        bn0.weight.normal_(1.0, 0.2)
        bn0.bias.normal_(0.0, 0.2)
    results = []
    for lean in (True, False):
        stageplan.ENABLED = lean
        try:
            conv, bn = copy.deepcopy(conv0), copy.deepcopy(bn0)
            outs = []
            for rep in range(2):
                x = x0.clone().requires_grad_(True)
                res = None
                kw = dict(stride=s, pad=p, transposed=tr, output_padding=op, training=True)
                z = A.conv_stage(x, conv, bn, getattr(L, act), getattr(C, pm), dtype, **kw)
                if use_res:
                    res = torch.randn(z.shape, generator=torch.Generator().manual_seed(rep)).to(dev).to(dtype).contiguous(
                        memory_format=torch.channels_last).requires_grad_(True)
                    z = A.conv_stage(x, conv, bn, getattr(L, act), getattr(C, pm), dtype, res1=res, **kw)
                g = torch.randn(z.shape, generator=torch.Generator().manual_seed(7 + rep)).to(dev).to(dtype).contiguous(
                    memory_format=torch.channels_last)
                z.backward(g)
                L.flush_bn_counters()
                outs += [z.detach().clone(), x.grad.clone()] + ([res.grad.clone()] if res is not None else [])
            outs += [conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_mean.clone(),
                     bn.running_var.clone(), bn.num_batches_tracked.clone()]
            assert conv.bias.grad is None or not conv.bias.grad.any()
            planned = any(v for v in conv.__dict__.get("_ir2rgb_plans", {}).values())
            assert planned == lean
            results.append(outs)
        finally:
            stageplan.ENABLED = True
    assert len(results[0]) == len(results[1])
    for i, (a, b) in enumerate(zip(*results)):
        assert torch.equal(a, b), f"output {i} differs between the planned and the general stage"
        assert torch.isfinite(a.float()).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # Cin, H, W, Cout, k, stride, pad, groups, group order, active groups in the flagged pass
    (64, 33, 65, 128, 4, 2, 2, 3, (2, 0, 1), 2),      # image discriminator layer: three groups, the generator's pass on two
    (128, 17, 33, 256, 4, 1, 2, 2, (1, 0), 1),        # temporal discriminator layer, stride 1: two groups, one active
    (64, 70, 200, 128, 4, 2, 2, 3, None, 2),          # > 128 statistics rows per group: finalize + apply launches
])
def test_planned_grouped_stage_equals_the_general_stage(dev, dtype, case):
    """stageplan.GroupedStagePlan (batched discriminator stages: BatchNorm per sample group, backward flags, passes with
    inactive groups) against ConvStageFn's general code: forward, a flagged pass (no parameter gradients, leading groups
    only) and a full pass (parameter gradients, accumulated in the kernel on top of an earlier one), bit for bit."""
    import copy
    from ir2rgb_amd import autograd as A, conv as C, layers as L, stageplan
    cin, h, w, cout, k, s, p, G, order, active = case
    gen = torch.Generator().manual_seed(cin + h + G)
    x0 = torch.randn(2 * G, cin, h, w, generator=gen).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    conv0 = torch.nn.Conv2d(cin, cout, k, s, p).to(dev)
    bn0 = torch.nn.BatchNorm2d(cout).to(dev)
    with torch.no_grad():
        bn0.weight.normal_(1.0, 0.2)
        bn0.bias.normal_(0.0, 0.2)
    results = []
    for lean in (True, False):
        stageplan.ENABLED = lean
        try:
            conv, bn = copy.deepcopy(conv0), copy.deepcopy(bn0)
            holder = torch.nn.Sequential(conv, bn)
            outs = []
            for rep in range(2):
                x = x0.clone().requires_grad_(True)
                with L.repeated_forward(tuple(range(1, G + 1))):
                    z = A.conv_stage(x, conv, bn, L.ACT_LEAKY, C.PAD_ZERO, dtype, groups=G, group_order=order, training=True)
                L.flush_bn_counters()
                g = torch.randn(z.shape, generator=torch.Generator().manual_seed(7 + rep)).to(dev).to(dtype).contiguous(
                    memory_format=torch.channels_last)
                na = z.shape[0] // G * active
                with A.backward_flags([holder], A.SKIP_PARAM_GRADS, active):
                    z.backward(g, retain_graph=True, inputs=[x])
                dx_part = x.grad[:na].clone()
                assert conv.weight.grad is None or rep > 0
                x.grad = None
                with A.backward_flags([holder], A.SKIP_INPUT_GRAD):
                    z.backward(g, inputs=[x, conv.weight, bn.weight, bn.bias])
                outs += [z.detach().clone(), dx_part, x.grad.clone()]
            outs += [conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_mean.clone(),
                     bn.running_var.clone(), bn.num_batches_tracked.clone()]
            planned = any(isinstance(v, stageplan.GroupedStagePlan) for v in conv.__dict__.get("_ir2rgb_plans", {}).values())
            assert planned == lean
            results.append(outs)
        finally:
            stageplan.ENABLED = True
    for i, (a, b) in enumerate(zip(*results)):
        assert torch.equal(a, b), f"output {i} differs between the planned and the general grouped stage"
        assert torch.isfinite(a.float()).all()


@pytest.mark.parametrize("case", [
    # Cin, W, KW, stride, pad, reflect
    (9, 300, 7, 1, 3, 1),       # the generators' 7-wide first layer: three 128-column tiles, ragged last one, both mirrors
    (16, 300, 4, 2, 2, 0),      # the discriminators' stride-2 first layer at the channel limit (16 x 4 = 64)
    (6, 257, 4, 2, 1, 0),       # odd width: a one-column last tile
    (6, 128, 3, 1, 1, 1),       # exactly one tile
    (3, 131, 7, 1, 3, 1),       # the right mirrors start in the tile BEFORE the last (3 columns wide)
    (5, 2, 3, 1, 1, 1),         # narrower than the kernel
])
def test_xexpand_bwd_row_tile_kernel_vs_autograd(dev, case):
    """ir2rgb_xexpand_bwd (rows of the expanded gradient staged once per 128-column tile) against autograd through a
    plain-torch restatement of the expansion (pad along x, unfold): fp32 sums of the same half values."""
    from ir2rgb_amd import autograd as A
    cin, w, kw, sx, px, reflect = case
    g = torch.Generator().manual_seed(31)
    n, h = 2, 5
    wout = (w + 2 * px - kw) // sx + 1
    d = torch.randn(n, 64, h, wout, generator=g).to(dev).to(torch.float16).contiguous(memory_format=torch.channels_last)
    img = torch.zeros(n, cin, h, w, device=dev, requires_grad=True)
    xp = F.pad(img, (px, px, 0, 0), mode="reflect" if reflect else "constant")
    e = xp.unfold(3, kw, sx)                                  # [n, cin, h, wout, kw]
    e = e.permute(0, 1, 4, 2, 3).reshape(n, cin * kw, h, wout)
    (e * d[:, :cin * kw].float()).sum().backward()
    got = A.xexpand_bwd(d, cin, w, kw, sx, px, reflect)
    assert got.shape == img.grad.shape
    torch.testing.assert_close(got, img.grad, atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # Cin, H, W, Cout, k, stride, pad, pad_mode, transposed, opad
    (64, 16, 64, 128, 3, 1, 1, 1, False, 0),      # one image row per K-step
    (192, 9, 21, 136, 3, 1, 1, 1, False, 0),      # ragged pixels, channels not multiples of 128
    (64, 33, 47, 64, 4, 2, 2, 0, False, 0),       # discriminator, odd sizes
    (128, 12, 20, 64, 3, 2, 1, 0, True, 1),       # transposed
    (64, 40, 72, 256, 7, 1, 3, 1, False, 0),      # 49 taps
    (1024, 32, 64, 1024, 3, 1, 1, 1, False, 0),   # bottleneck of the 256x512 generator: nine-tap kernel, no split
    (64, 8, 128, 64, 3, 1, 1, 0, False, 0),       # nine-tap kernel, zero padding, two segments per row, split-K
    (128, 5, 64, 192, 3, 1, 1, 0, False, 0),      # nine-tap kernel, 3 x 2 tiles, odd row count
    (128, 6, 192, 64, 3, 1, 1, 1, False, 0),      # nine-tap kernel, reflect, three segments per row
    (64, 33, 47, 128, 3, 2, 1, 0, False, 0),      # stride-2 3x3 (parity-split patch), odd sizes, one ragged segment
    (64, 20, 300, 64, 3, 2, 1, 0, False, 0),      # ... three segments per output row, the last one ragged
    (128, 9, 70, 64, 3, 2, 1, 0, True, 1),        # ... transposed, two segments per input row
    (256, 16, 32, 128, 3, 2, 1, 0, False, 0),     # ... 2 x 4 channel tiles
])
def test_wgrad_kernel_vs_torch(dev, dtype, case):
    """MFMA weight-gradient kernel alone (transposed LDS reads, split-K atomics) vs torch's fp32
    backward-filter on the same half-rounded operands.  Tolerance: relative L2 <= 2e-3 (fp32
    accumulation of exactly-representable products; only the summation order differs)."""
    from ir2rgb_amd import conv as C
    cin, h, w, cout, k, s, p, pm, tr, op = case
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(2 if cin < 1024 else 1, cin, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    desc = C.make_desc(tuple(x.shape), cout, k, s, p, pm, dtype, tr, op)
    gy = torch.randn(x.shape[0], cout, desc.Hout, desc.Wout, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    dw = C.conv2d_wgrad(desc, x, gy)
    xr = x.float()
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    wr = torch.zeros(wshape, device=dev, requires_grad=True)
    if tr:
        y = F.conv_transpose2d(xr, wr, None, s, p, op)
    elif pm:
        y = F.conv2d(F.pad(xr, (p,) * 4, mode="reflect"), wr, None, s)
    else:
        y = F.conv2d(xr, wr, None, s, p)
    (y * gy.float()).sum().backward()
    assert dw.shape == wr.grad.shape
    assert _rel(dw, wr.grad) <= 2e-3, _rel(dw, wr.grad)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # N, Cin, H, W, Cout, (kh, kw), (sh, sw), (ph, pw), reflect, accumulate
    (2, 64, 21, 130, 64, (7, 1), (1, 1), (3, 0), True, False),     # first-layer 7x1 pass: reflect along y, ragged third segment
    (1, 64, 40, 64, 64, (7, 1), (1, 1), (3, 0), True, True),       # dw += (a second use of the weight)
    (1, 64, 24, 64, 128, (4, 1), (2, 1), (2, 0), False, False),    # two a-tiles
    (2, 64, 9, 200, 64, (1, 7), (1, 1), (0, 3), True, False),      # head 1x7 pass: reflect along x across segment borders
    (1, 128, 12, 128, 64, (1, 7), (1, 1), (0, 3), True, False),    # two b-tiles (128-channel feature map)
    (3, 64, 37, 65, 64, (4, 1), (2, 1), (2, 0), False, False),     # discriminator 4x1 stride 2, zero padding, odd sizes, 3 samples
    (2, 64, 512, 66, 64, (4, 1), (2, 1), (2, 0), False, True),     # many K-steps: 256 splits, wide finish pass, accumulate
    (1, 64, 6, 10, 64, (7, 1), (1, 1), (3, 0), True, False),       # fewer K-steps than splits could hold
])
def test_line_wgrad_kernel_vs_torch(dev, dtype, case):
    """conv_wgrad_line_kernel (all taps of a k x 1 / 1 x k layer from one staged tile per 64 x 64 channel tile, per-split
    slabs summed in a fixed order) vs torch's fp32 backward-filter on the same half-rounded operands, relative L2 <= 2e-3;
    two calls agree bit for bit (no atomics)."""
    from ir2rgb_amd import conv as C
    n, cin, h, w, cout, k, s, p, reflect, accumulate = case
    g = torch.Generator().manual_seed(cin + h + w)
    x = torch.randn(n, cin, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    desc = C.make_desc(tuple(x.shape), cout, k, s, p, C.PAD_REFLECT if reflect else C.PAD_ZERO, dtype)
    gy = torch.randn(n, cout, desc.Hout, desc.Wout, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    wr = torch.zeros((cout, cin) + k, device=dev, requires_grad=True)
    xr = x.float()
    if reflect:
        xr = F.pad(xr, (p[1], p[1], p[0], p[0]), mode="reflect")
        y = F.conv2d(xr, wr, None, s)
    else:
        y = F.conv2d(xr, wr, None, s, p)
    (y * gy.float()).sum().backward()
    if accumulate:
        base = torch.randn(wr.shape, generator=g).to(dev)
        dw = base.clone()
        C.conv2d_wgrad(desc, x, gy, out=dw, accumulate=True)
        dw2 = base.clone()
        C.conv2d_wgrad(desc, x, gy, out=dw2, accumulate=True)
        want = base + wr.grad
    else:
        dw, dw2, want = C.conv2d_wgrad(desc, x, gy), C.conv2d_wgrad(desc, x, gy), wr.grad
    assert dw.shape == want.shape and torch.equal(dw, dw2)
    assert _rel(dw - (base if accumulate else 0), wr.grad) <= 2e-3, _rel(dw - (base if accumulate else 0), wr.grad)
    assert torch.allclose(dw, want, rtol=1e-2, atol=1e-2 * float(wr.grad.abs().max()))


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-2), (torch.bfloat16, 4e-2)])
def test_head_and_warp_blend_backward(dev, dtype, tol):
    """Separable 7x7 heads (tanh image head; flow*20 + sigmoid weight heads sharing one feature map) and
    the warp-blend, forward and backward, vs torch fp32 autograd of the reference formulas."""
    from ir2rgb_amd import autograd as A
    from ir2rgb_amd.networks import get_grid
    g = torch.Generator().manual_seed(31)
    n, c, h, w = 1, 64, 24, 40
    feat0 = torch.randn(n, c, h, w, generator=g).to(dev).to(dtype)
    conv_f, conv_w = nn.Conv2d(c, 2, 7).to(dev), nn.Conv2d(c, 1, 7).to(dev)
    conv_i = nn.Conv2d(c, 3, 7).to(dev)
    with torch.no_grad():
        for cv, sc in ((conv_f, 0.002), (conv_w, 0.02), (conv_i, 0.02)):
            cv.weight.copy_((torch.randn(cv.weight.shape, generator=g) * sc).to(dtype).float())
    prev = torch.rand(n, 6, h, w, generator=g).to(dev)
    proj = torch.randn(n, 3, h, w, generator=g).to(dev)

    def ref():
        f = _leaf(feat0.float())
        ws = [_leaf(cv.weight) for cv in (conv_f, conv_w, conv_i)]
        bs = [_leaf(cv.bias) for cv in (conv_f, conv_w, conv_i)]
        fp = F.pad(f, (3,) * 4, mode="reflect")
        flow = F.conv2d(fp, ws[0], bs[0]) * 20
        wgt = torch.sigmoid(F.conv2d(fp, ws[1], bs[1]))
        raw = torch.tanh(F.conv2d(fp, ws[2], bs[2]))
        grid = get_grid(n, h, w, device=dev)
        fl = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], 1)
        warp = F.grid_sample(prev[:, -3:], (grid + fl).permute(0, 2, 3, 1), mode="bilinear", padding_mode="border", align_corners=False)
        out = raw * wgt + warp * (1 - wgt)
        ((out * proj).sum() + flow.pow(2).mean() + wgt.sum() * 0.1).backward()
        return out, f.grad, [x.grad for x in ws], [x.grad for x in bs]

    out_r, gf_r, gw_r, gb_r = ref()
    fh = feat0.contiguous(memory_format=torch.channels_last).requires_grad_()
    fw = A.head_stage(fh, [conv_f, conv_w], [0, 0, 2], mul=20.0)
    raw = A.head_stage(fh, [conv_i], [1, 1, 1])
    flow, wgt = fw[:, 0:2], fw[:, 2:3]
    out_h = A.warp_blend(raw, prev, flow, wgt)
    assert _rel(out_h, out_r) <= tol
    ((out_h * proj).sum() + flow.pow(2).mean() + wgt.sum() * 0.1).backward()
    errs = {"dfeat": _rel(fh.grad, gf_r)}
    for name, cv, gw, gb in zip(("flow", "w", "img"), (conv_f, conv_w, conv_i), gw_r, gb_r):
        errs[f"dW_{name}"] = _rel(cv.weight.grad, gw)
        errs[f"db_{name}"] = _rel(cv.bias.grad, gb)
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, errs


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.5e-2)])
@pytest.mark.parametrize("C,H,W,act", [(1024, 32, 64, 1), (512, 61, 67, 2), (256, 33, 65, 2), (64, 3, 5, 1), (128, 64, 64, 1), (2048, 4, 8, 0)])
def test_bn_backward_kernels_vs_autograd(dev, dtype, tol, C, H, W, act):
    """ir2rgb_bn_bwd against torch autograd of act(batch_norm(y)) in fp32 on the same half-rounded y and gz: the one-launch
    form (a workgroup owns 8-32 channels over all pixels, slabs in registers: every case here with <= 4096 pixels,
    K = 4 and K = 8 variants, all three channel-group widths) and the two-launch form (128 x 64x64), plus the in-kernel
    accumulation of dgamma / dbeta over a second call (sample groups)."""
    from ir2rgb_amd import autograd as A
    from ir2rgb_amd import layers as L
    g = torch.Generator().manual_seed(C + H)
    y = (torch.randn(1, C, H, W, generator=g) * 1.5 + 0.3).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    gz = torch.randn(1, C, H, W, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
    yr = y.float().contiguous().detach().requires_grad_()
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    with torch.backends.cudnn.flags(enabled=False):      # (torch's own kernels: MIOpen's NHWC BatchNorm crashed on one of these shapes)
        z = F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5)
        z = F.relu(z) if act == 1 else (F.leaky_relu(z, 0.2) if act == 2 else z)
        z.backward(gz.float().contiguous())
    mean = y.float().mean((0, 2, 3))
    invstd = (y.float().var((0, 2, 3), unbiased=False) + 1e-5).rsqrt()
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    gy, dgamma, dbeta = A.bn_bwd(gz, y, scale, shift, mean, invstd, act)
    assert _rel(gy, yr.grad) <= tol, _rel(gy, yr.grad)
    assert _rel(dgamma, gr.grad) <= 1e-3 and _rel(dbeta, br.grad) <= 1e-3
    # a second sample group accumulating into the same parameter gradients
    _, dg2, db2 = A.bn_bwd(gz, y, scale, shift, mean, invstd, act, params=(dgamma.clone(), dbeta.clone()))
    torch.testing.assert_close(dg2, 2 * dgamma, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(db2, 2 * dbeta, rtol=1e-6, atol=1e-6)
