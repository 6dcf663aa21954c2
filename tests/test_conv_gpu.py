"""GPU parity tests of the MFMA implicit-GEMM convolution (through the C ABI) against torch
fp32 convolutions evaluated on the SAME half-rounded inputs and weights, so that the only
differences are fp32 accumulation order and the final rounding of the output to half.

Tolerance: |delta| <= 2^-8 * |ref| + 1e-2 * rms(ref) for bf16 (one bf16 ulp of the output plus
accumulation noise), 2^-11 * |ref| + 2e-3 * rms(ref) for f16.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_and_run(dev, dtype, N, Cin, H, W, Cout, k, stride, pad, pad_mode, transposed=False, opad=0, act=0, seed=0):
    from ir2rgb_amd import conv as C
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    w = (torch.randn(wshape, generator=g) * (1.0 / np.sqrt(Cin * k * k))).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    desc = C.make_desc(x.shape, Cout, k, stride, pad, pad_mode, dtype, transposed, opad, act)
    wp = C.pack_weight(desc, w)
    y, stats = C.conv2d_fwd(desc, x, wp, b, want_stats=True)
    xr, wr = x.float(), w.to(dtype).float()
    if transposed:
        ref = F.conv_transpose2d(xr, wr, b, stride=stride, padding=pad, output_padding=opad)
    else:
        if pad_mode == 1 and pad > 0:
            xr = F.pad(xr, (pad,) * 4, mode="reflect")
            ref = F.conv2d(xr, wr, b, stride=stride)
        else:
            ref = F.conv2d(xr, wr, b, stride=stride, padding=pad)
    if act == 1:
        ref = F.leaky_relu(ref, 0.2)
    return y, stats, ref


def _check(y, stats, ref, dtype):
    assert y.shape == ref.shape
    yf = y.float()
    rms = ref.pow(2).mean().sqrt().item()
    rel, ab = (2.0 ** -8, 1e-2) if dtype == torch.bfloat16 else (2.0 ** -11, 2e-3)
    err = (yf - ref).abs()
    bound = rel * ref.abs() + ab * rms
    bad = (err > bound).sum().item()
    assert bad == 0, f"{bad} elements out of tolerance; max err {err.max().item():.4g}, rms {rms:.4g}"
    # BN statistics: per-channel sum / sum of squares over all pixels of the fp32 result
    s = stats.sum(0)
    n = ref.numel() / ref.shape[1]
    torch.testing.assert_close(s[0] / n, ref.mean((0, 2, 3)), atol=2e-3 * max(rms, 1e-3), rtol=1e-3)
    torch.testing.assert_close(s[1] / n, ref.pow(2).mean((0, 2, 3)), atol=2e-3 * max(rms * rms, 1e-3), rtol=2e-3)


CASES = [
    # N, Cin, H, W, Cout, k, stride, pad, pad_mode, transposed, opad
    (1, 64, 16, 32, 128, 3, 1, 1, 1, False, 0),      # ResnetBlock-style reflect 3x3, TP=64
    (1, 128, 32, 64, 128, 3, 1, 1, 1, False, 0),     # TP=64, two K chunks
    (2, 64, 24, 40, 192, 3, 1, 1, 1, False, 0),      # batch 2, Cout not a multiple of 128
    (1, 64, 33, 47, 64, 3, 2, 1, 0, False, 0),       # stride 2 zero pad, odd sizes, Cout 64
    (1, 64, 20, 36, 128, 4, 2, 2, 0, False, 0),      # discriminator 4x4 s2 p2
    (1, 128, 17, 19, 8, 4, 1, 2, 0, False, 0),       # discriminator 4x4 s1 p2, tiny Cout
    (1, 128, 16, 24, 64, 3, 2, 1, 0, True, 1),       # ConvTranspose 3x3 s2 p1 op1
    (1, 64, 9, 13, 32, 4, 2, 1, 0, True, 0),         # ConvTranspose 4x4 s2 p1 (FlowNet deconv)
    (1, 64, 64, 128, 128, 3, 1, 1, 1, False, 0),     # 8192 pixels -> TP=128 path
    (1, 64, 128, 256, 128, 3, 1, 1, 1, False, 0),    # 32768 pixels -> TP=256 path
    (1, 64, 12, 20, 130, 1, 1, 0, 0, False, 0),      # 1x1 (runtime tap path), ragged Cout
    (1, 64, 40, 24, 128, 7, 1, 3, 1, False, 0),      # 7x7 reflect (runtime tap path, 49 taps)
    # patch-staged 3x3 kernel (conv3x3_patch.hip): Cin >= 256, >= 200 tiles
    (1, 256, 101, 250, 256, 3, 1, 1, 1, False, 0),   # 2x128 px x 128 cout tiles, reflect, ragged rows / columns
    (1, 256, 64, 126, 1024, 3, 1, 2, 0, False, 0),   # data-gradient geometry: zero pad 2, output 66x128
    (1, 320, 67, 128, 192, 3, 1, 1, 0, False, 0),    # 2x64 px x 64 cout tiles (Cout % 128 != 0), zero pad, 5 K chunks
    (2, 256, 34, 64, 320, 3, 1, 1, 1, False, 0),     # 2x64 tiles, batch 2, reflect
    # ... its split-K form (two workgroups per 2x64 px x 128 cout tile, Cin >= 512)
    (1, 512, 52, 128, 256, 3, 1, 1, 1, False, 0),    # reflect, 4 + 4 K chunks
    (2, 640, 35, 64, 384, 3, 1, 1, 0, False, 0),     # zero pad, batch 2, 5 + 5 K chunks, ragged last row
]
SPLIT_CASES = CASES[-2:]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_conv_vs_torch(dev, dtype, case):
    N, Cin, H, W, Cout, k, s, p, pm, tr, op = case
    y, stats, ref = _ref_and_run(dev, dtype, N, Cin, H, W, Cout, k, s, p, pm, tr, op, seed=Cin + H)
    _check(y, stats, ref, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(3, 512, 35, 67, 4, 1, 2), (1, 512, 9, 13, 4, 1, 2), (2, 1024, 12, 20, 3, 2, 1),
                                  (1, 512, 8, 8, 1, 1, 0)])
def test_one_channel_fp32_output(dev, dtype, case):
    """The PatchGAN logit layers (512 -> 1, 4x4, zero padding, fp32 output): conv_dot_kernel (a wave per output
    pixel) against torch's convolution on the same half-rounded operands; fp32 accumulation in another order."""
    from ir2rgb_amd import conv as C
    N, Cin, H, W, k, stride, pad = case
    g = torch.Generator(device="cpu").manual_seed(Cin + H)
    x = torch.randn(N, Cin, H, W, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(1, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)).to(dev)
    b = torch.randn(1, generator=g).to(dev)
    desc = C.make_desc(x.shape, 1, k, stride, pad, C.PAD_ZERO, dtype, out_f32=True)
    if os.environ.get("IR2RGB_CONV_DOT", "1") != "0":
        assert C.kernel_name(desc) == "conv_dot_kernel"
    y, _ = C.conv2d_fwd(desc, x, C.pack_weight(desc, w), b)
    ref = F.conv2d(x.float(), w.to(dtype).float(), b, stride=stride, padding=pad)
    assert y.dtype == torch.float32 and y.shape == ref.shape
    rms = ref.pow(2).mean().sqrt().item()
    assert (y - ref).abs().max().item() <= 1e-4 * max(rms, 1e-3) + 1e-5


def test_conv_fused_leaky(dev):
    y, stats, ref = _ref_and_run(dev, torch.bfloat16, 1, 64, 20, 36, 64, 4, 2, 2, 0, act=1)
    _check(y, stats, ref, torch.bfloat16)


def test_conv_exact_integers(dev):
    """Small-integer data is exact in bf16 and in fp32 accumulation: bit-exact against torch.
    Asymmetric data catches transposed fragment maps."""
    from ir2rgb_amd import conv as C
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randint(-2, 3, (1, 64, 10, 12), generator=g).float().to(dev).to(torch.bfloat16)
    x = x.contiguous(memory_format=torch.channels_last)
    w = torch.randint(-2, 3, (128, 64, 3, 3), generator=g).float().to(dev)
    desc = C.make_desc(x.shape, 128, 3, 1, 1, 1, torch.bfloat16)
    y, _ = C.conv2d_fwd(desc, x, C.pack_weight(desc, w))
    ref = F.conv2d(F.pad(x.float(), (1,) * 4, mode="reflect"), w)
    assert ref.abs().max() < 256  # exactly representable in bf16? integers up to 256 are
    assert torch.equal(y.float(), ref)


def test_hot_shape_linearity(dev):
    """Full-size bottleneck conv of the 512x1024 generator (1024->1024 @64x128): linearity in the
    input and agreement with torch on a random subset of output channels."""
    from ir2rgb_amd import conv as C
    g = torch.Generator(device="cpu").manual_seed(5)
    x1 = torch.randn(1, 1024, 64, 128, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(1024, 1024, 3, 3, generator=g) * 0.01).to(dev)
    desc = C.make_desc(x1.shape, 1024, 3, 1, 1, 1, torch.bfloat16)
    wp = C.pack_weight(desc, w)
    y1, stats = C.conv2d_fwd(desc, x1, wp, want_stats=True)
    y2, _ = C.conv2d_fwd(desc, (x1 * 2).contiguous(memory_format=torch.channels_last), wp)
    assert torch.equal(y2.float(), y1.float() * 2)  # scaling by 2 is exact in bf16/fp32
    idx = torch.tensor([0, 1, 127, 128, 511, 777, 1023], device=dev)
    ref = F.conv2d(F.pad(x1.float(), (1,) * 4, mode="reflect"), w[idx].to(torch.bfloat16).float())
    got = y1.float()[:, idx]
    rms = ref.pow(2).mean().sqrt().item()
    assert ((got - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-2 * rms).all()
    torch.testing.assert_close(stats.sum(0)[0][idx] / 8192, ref.mean((0, 2, 3)), atol=2e-3 * rms, rtol=1e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # N, forward Cout (= channels of gy), H, W, forward Cin
    (1, 256, 100, 128, 256),     # 2x64 px x 64 cout tiles: every tile touches the left and right border
    (1, 256, 100, 256, 256),     # 2x128 px x 128 cout tiles
    (2, 320, 52, 64, 256),       # one column tile (left and right border in the same tile), batch 2, 5 K chunks
    (1, 512, 52, 128, 512),      # split-K form: 2x64 px x 128 cout tiles, two workgroups each
    (2, 640, 52, 64, 256),       # split-K form, one column tile, batch 2, 5 + 5 K chunks
])
def test_reflect_adjoint_in_place(dev, dtype, case):
    """pad_mode 2 of the patch-staged kernel: the data gradient of a reflection-padded 3x3 convolution computed
    on the unpadded grid (border terms inside the kernel) vs torch autograd through F.pad(reflect) + conv2d on the
    same half-rounded operands."""
    from ir2rgb_amd import conv as C
    if os.environ.get("IR2RGB_CONV3X3P") == "0":
        pytest.skip("patch-staged kernel disabled by IR2RGB_CONV3X3P=0")
    n, cgy, h, w, cin = case
    g = torch.Generator(device="cpu").manual_seed(cgy + h)
    gy = torch.randn(n, cgy, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cgy, cin, 3, 3, generator=g) * (1.0 / np.sqrt(cgy * 9))).to(dev)
    dadj = C.make_desc(gy.shape, cin, 3, 1, 1, C.PAD_REFLECT_ADJ, dtype)
    assert C.kernel_name(dadj) == "conv3x3_patch_kernel"
    wp = C.pack_weight(C.make_desc(gy.shape, cin, 3, 1, 1, C.PAD_ZERO, dtype), wt, adjoint=True)
    dx, _ = C.conv2d_fwd(dadj, gy, wp)
    x = torch.zeros(n, cin, h, w, device=dev, requires_grad=True)
    y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), wt.to(dtype).float())
    (y * gy.float()).sum().backward()
    ref = x.grad
    rms = ref.pow(2).mean().sqrt().item()
    rel, ab = (2.0 ** -8, 1e-2) if dtype == torch.bfloat16 else (2.0 ** -11, 2e-3)
    err = (dx.float() - ref).abs()
    bad = (err > rel * ref.abs() + ab * rms).sum().item()
    assert bad == 0, f"{bad} elements out of tolerance; max err {err.max().item():.4g}, rms {rms:.4g}"
    # the general kernel refuses the mode instead of computing something else
    small = C.make_desc((1, 64, 8, 64), 64, 3, 1, 1, C.PAD_REFLECT_ADJ, dtype)
    assert C.kernel_name(small) == ""


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", SPLIT_CASES)
def test_split_k_form(dev, dtype, case):
    """The split-K form of the patch-staged kernel (ir2rgb_conv2d_fwd_ws with a workspace): selected for these shapes,
    bit-identical from launch to launch (the two partial tiles are added in whichever order the workgroups finish:
    a + b == b + a), BatchNorm partial sums included, and equal to the unsplit form (ir2rgb_conv2d_fwd: no workspace)
    up to the fp32 summation order."""
    import ctypes
    from ir2rgb_amd import conv as C, _lib
    if os.environ.get("IR2RGB_CONV3X3P") == "0" or os.environ.get("IR2RGB_CONV3X3P_SPLIT") == "0":
        pytest.skip("split-K form disabled by the environment")
    N, Cin, H, W, Cout, k, stride, pad, pad_mode = case[:9]
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(N, Cin, H, W, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / np.sqrt(Cin * k * k))).to(dev)
    desc = C.make_desc(x.shape, Cout, k, stride, pad, pad_mode, dtype)
    assert C.kernel_name(desc) == "conv3x3_patch_kernel"
    assert _lib.lib().ir2rgb_conv2d_fwd_workspace_bytes(ctypes.byref(desc)) > 0
    wp = C.pack_weight(desc, w)
    y0, s0 = C.conv2d_fwd(desc, x, wp, want_stats=True)
    for _ in range(10):
        y, s = C.conv2d_fwd(desc, x, wp, want_stats=True)
        assert torch.equal(y, y0) and torch.equal(s, s0)
    # the same workspace serves another shape in between (tickets advance by a fixed amount per launch)
    other = SPLIT_CASES[0] if case is SPLIT_CASES[1] else SPLIT_CASES[1]
    _ref_and_run(dev, dtype, *other)
    y, s = C.conv2d_fwd(desc, x, wp, want_stats=True)
    assert torch.equal(y, y0) and torch.equal(s, s0)
    # unsplit form through the plain entry point
    y1 = torch.empty_like(y0)
    s1 = torch.empty_like(s0)
    rc = _lib.lib().ir2rgb_conv2d_fwd(ctypes.byref(desc), C._p(x), C._p(wp), None, C._p(y1), C._p(s1), _lib.current_stream(x))
    assert rc == 0
    rms = y1.float().pow(2).mean().sqrt().item()
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10         # spacing of the half grid, relative, at most
    d = (y0.float() - y1.float()).abs()
    assert (d <= ulp * y1.float().abs() + 2e-5 * rms).all()          # the last bit of the half output (+ fp32 summation noise near 0)
    assert (d > 0).float().mean().item() < 0.05                        # ... and rarely
    torch.testing.assert_close(s0.sum(0), s1.sum(0), rtol=1e-5, atol=1e-4 * rms * rms * H * W)


def test_full_size_adjoint_identities(dev):
    """BASELINE config-2 bottleneck layer at full size (1024 -> 1024 channels, 3x3 reflect, 32x64 pixels), checked
    through the adjoint identities that tie the three kernels of a layer together without a CPU-sized oracle:
        <conv(x; W), g>  ==  <x, dgrad(g; W)>  ==  <W, wgrad(x, g)>
    with conv = patch-staged forward, dgrad = its in-place reflection adjoint (pad_mode 2), wgrad = the nine-tap
    weight-gradient kernel.  The three sums of 2M..9M products agree to the rounding of the half outputs."""
    from ir2rgb_amd import conv as C
    if os.environ.get("IR2RGB_CONV3X3P") == "0":
        pytest.skip("patch-staged kernel disabled by IR2RGB_CONV3X3P=0")
    dt = torch.bfloat16
    gen = torch.Generator(device="cpu").manual_seed(21)
    x = torch.randn(1, 1024, 32, 64, generator=gen).to(dev).to(dt).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(1, 1024, 32, 64, generator=gen).to(dev).to(dt).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(1024, 1024, 3, 3, generator=gen) * 0.01).to(dev).to(dt).float()    # exactly representable weights
    dfwd = C.make_desc(x.shape, 1024, 3, 1, 1, C.PAD_REFLECT, dt)
    assert C.kernel_name(dfwd) == "conv3x3_patch_kernel"
    y, _ = C.conv2d_fwd(dfwd, x, C.pack_weight(dfwd, w))
    dadj = C.make_desc(gy.shape, 1024, 3, 1, 1, C.PAD_REFLECT_ADJ, dt)
    assert C.kernel_name(dadj) == "conv3x3_patch_kernel"
    dx, _ = C.conv2d_fwd(dadj, gy, C.pack_weight(C.make_desc(gy.shape, 1024, 3, 1, 1, C.PAD_ZERO, dt), w, adjoint=True))
    dw = C.conv2d_wgrad(dfwd, x, gy)
    a = (y.double() * gy.double()).sum().item()
    b = (x.double() * dx.double()).sum().item()
    c = (w.double() * dw.double()).sum().item()
    scale = (y.double().pow(2).sum().sqrt() * gy.double().pow(2).sum().sqrt()).item()   # Cauchy-Schwarz scale of the sums
    # y and dx are rounded to bf16 (2^-9 relative per element, random signs over 2M terms); dw is fp32
    assert abs(a - c) <= 2e-4 * scale, (a, c, scale)
    assert abs(b - c) <= 2e-4 * scale, (b, c, scale)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,Cin,H,W,Cout", [(1, 128, 9, 300, 24), (2, 64, 5, 129, 24), (1, 128, 3, 70, 21), (1, 64, 4, 1024, 24),
                                            (1, 128, 2, 5, 24)])
def test_head_1x7_pass_row_segment_kernel(dev, dtype, N, Cin, H, W, Cout):
    """conv1x7_thin_kernel (the 1x7 pass of the separable heads, reference networks.py:165-171 as layers.head_stage
    evaluates them: reflection pad 3, Cout*7 <= 24 fp32 row responses, no bias): against torch's fp32 convolution on the
    same half-rounded operands -- ragged widths (last segment mostly empty, images narrower than a segment), both channel
    counts, a Cout that is not a multiple of 4, operands inside NaN guard bands and the output inside canary bands."""
    from ir2rgb_amd import conv as C
    from test_bounds_gpu import Guarded
    g = torch.Generator(device="cpu").manual_seed(Cin + W)
    x = torch.randn(N, Cin, H, W, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 1, 7, generator=g) * (1.0 / np.sqrt(Cin * 7))).to(dev)
    desc = C.make_desc(tuple(x.shape), Cout, (1, 7), 1, (0, 3), C.PAD_REFLECT, dtype, out_f32=True)
    wp = C.pack_weight(desc, w)
    if Cout % 4 == 0:
        assert C.kernel_name(desc) == "conv1x7_thin_kernel"
    gx, gw = Guarded(x, float("nan")), Guarded(wp, float("nan"))
    y0, _ = C.conv2d_fwd(desc, x, wp)
    gy = Guarded(y0, 7.0)
    gy.t.zero_()
    y1, _ = C.conv2d_fwd(desc, gx.t, gw.t, out=gy.t)
    torch.cuda.synchronize()
    assert torch.equal(y1, y0) and gy.intact() and gx.intact() and gw.intact()
    ref = F.conv2d(F.pad(x.float(), (3, 3, 0, 0), mode="reflect"), w.to(dtype).float())
    assert y0.dtype == torch.float32 and y0.shape == ref.shape
    rms = ref.pow(2).mean().sqrt().item()
    assert ((y0 - ref).abs() <= 1e-5 * ref.abs() + 2e-4 * rms).all(), (y0 - ref).abs().max().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W,Cout", [(1, 40, 100, 128), (2, 9, 33, 64), (1, 64, 128, 64), (1, 5, 31, 128), (1, 256, 96, 128)])
def test_first_layer_7x1_pass_column_tile_kernel(dev, dtype, N, H, W, Cout):
    """conv7x1_col_kernel (the 7x1 pass of the generators' first layers, reference networks.py:141,:150,:253-255 as
    layers.first_stage evaluates them: 64 x-expanded channels, reflection pad 3 in y): against torch's fp32 convolution on
    the same half-rounded operands, with bias, with BatchNorm partial sums; ragged tiles on both axes, images lower than a
    tile, more tiles than persistent workgroups (256 x 96: 96 tiles... and 40 x 100), operands in NaN bands, outputs in canary bands."""
    from ir2rgb_amd import conv as C
    from test_bounds_gpu import Guarded
    g = torch.Generator(device="cpu").manual_seed(H + W)
    x = torch.randn(N, 64, H, W, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, 64, 7, 1, generator=g) * (1.0 / np.sqrt(64 * 7))).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    desc = C.make_desc(tuple(x.shape), Cout, (7, 1), 1, (3, 0), C.PAD_REFLECT, dtype)
    assert C.kernel_name(desc) == "conv7x1_col_kernel"
    wp = C.pack_weight(desc, w)
    y0, s0 = C.conv2d_fwd(desc, x, wp, b, want_stats=True)
    gx, gw, gb = Guarded(x, float("nan")), Guarded(wp, float("nan")), Guarded(b, float("nan"))
    gy, gs = Guarded(y0, 7.0), Guarded(s0, 7.0)
    gy.t.zero_()
    y1, _ = C.conv2d_fwd(desc, gx.t, gw.t, gb.t, want_stats=False, out=gy.t)
    torch.cuda.synchronize()
    assert torch.equal(y1, y0) and gy.intact() and gx.intact() and gw.intact() and gb.intact()
    ref = F.conv2d(F.pad(x.float(), (0, 0, 3, 3), mode="reflect"), w.to(dtype).float(), b)
    _check(y0, s0, ref, dtype)
