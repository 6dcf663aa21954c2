"""GPU parity tests: the HIP operators (through the C ABI) vs the CPU oracle, the golden
vectors, and size-independent properties at BASELINE.json's full sizes.

Tolerance: fp32 ops, max|delta| <= 1e-5 (only the summation order differs);
relative to max|ref| where magnitudes exceed 1.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu

ATOL = 1e-5


def _mods():
    from ir2rgb_amd.flownet2_pytorch.networks.channelnorm_package.channelnorm import ChannelNorm
    from ir2rgb_amd.flownet2_pytorch.networks.correlation_package.correlation import Correlation
    from ir2rgb_amd.flownet2_pytorch.networks.resample2d_package.resample2d import Resample2d
    return Correlation, Resample2d, ChannelNorm


def close(a, ref, atol=ATOL):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(a - ref).max())
    assert err <= atol * scale, f"max|delta| {err:.3e} > {atol * scale:.3e}"


def cu(a, dev, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(grad)


# ---------------------------------------------------------------- golden vectors
def test_golden_correlation(dev, golden_dir):
    Correlation, _, _ = _mods()
    for f in sorted(glob.glob(os.path.join(golden_dir, "ops_corr_*.npz"))):
        g = np.load(f)
        hp = [int(v) for v in g["hp"]]
        f1, f2 = cu(g["f1"], dev, True), cu(g["f2"], dev, True)
        out = Correlation(*hp, 1)(f1, f2)
        close(out, g["out"])
        out.backward(cu(g["gout"], dev))
        close(f1.grad, g["g1"])
        close(f2.grad, g["g2"])


def test_golden_resample(dev, golden_dir):
    _, Resample2d, _ = _mods()
    for f in sorted(glob.glob(os.path.join(golden_dir, "ops_resample_*.npz"))):
        g = np.load(f)
        img, flow = cu(g["img"], dev, True), cu(g["flow"], dev, True)
        out = Resample2d()(img, flow)
        close(out, g["out"])
        out.backward(cu(g["gout"], dev))
        close(img.grad, g["gimg"])
        close(flow.grad, g["gflow"], atol=5e-5)


def test_golden_channelnorm(dev, golden_dir):
    _, _, ChannelNorm = _mods()
    for f in sorted(glob.glob(os.path.join(golden_dir, "ops_cnorm_*.npz"))):
        g = np.load(f)
        x = cu(g["x"], dev, True)
        out = ChannelNorm()(x)
        close(out, g["out"], atol=1e-6)
        out.backward(cu(g["gout"], dev))
        close(x.grad, g["gin"])


# ---------------------------------------------------------------- seeded vs oracle
@pytest.mark.parametrize("N,C,H,W", [(1, 64, 16, 24), (2, 13, 9, 40), (1, 256, 8, 128), (1, 32, 5, 136), (1, 5, 3, 7)])
def test_correlation_vs_oracle_flownetc_config(dev, N, C, H, W):
    """k=1,s1=1,pad=md=20,s2=2: W % 8 == 0 takes the register-tiled wave kernel (incl. a
    second, partial x chunk at W=136), anything else the generic kernel."""
    Correlation, _, _ = _mods()
    rng = np.random.default_rng(N * 1000 + C + W)
    f1 = rng.standard_normal((N, C, H, W)).astype(np.float32)
    f2 = rng.standard_normal((N, C, H, W)).astype(np.float32)
    hp = (20, 1, 20, 1, 2)
    t1, t2 = cu(f1, dev, True), cu(f2, dev, True)
    out = Correlation(*hp, 1)(t1, t2)
    ref = O.correlation_fwd(f1, f2, *hp)
    assert tuple(out.shape) == ref.shape
    close(out, ref)
    go = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(cu(go, dev))
    g1, g2 = O.correlation_bwd(f1, f2, go, *hp)
    close(t1.grad, g1)
    close(t2.grad, g2)


@pytest.mark.parametrize("hp", [(4, 1, 4, 1, 1), (21, 3, 20, 2, 3), (3, 3, 2, 1, 2), (8, 1, 8, 2, 2), (20, 1, 20, 1, 4)])
def test_correlation_vs_oracle_other_params(dev, hp):
    """kernel_size 3, stride1 2, max_displacement not divisible by stride2, pad != md."""
    Correlation, _, _ = _mods()
    rng = np.random.default_rng(sum(hp))
    f1 = rng.standard_normal((2, 6, 20, 24)).astype(np.float32)
    f2 = rng.standard_normal((2, 6, 20, 24)).astype(np.float32)
    out = Correlation(*hp, 1)(cu(f1, dev), cu(f2, dev))
    ref = O.correlation_fwd(f1, f2, *hp)
    assert tuple(out.shape) == ref.shape
    close(out, ref)
    if hp[3] == 1:  # backward is defined for stride1 == 1 only
        go = rng.standard_normal(ref.shape).astype(np.float32)
        t1, t2 = cu(f1, dev, True), cu(f2, dev, True)
        Correlation(*hp, 1)(t1, t2).backward(cu(go, dev))
        g1, g2 = O.correlation_bwd(f1, f2, go, *hp)
        close(t1.grad, g1)
        close(t2.grad, g2)


def test_correlation_backward_stride1_not_one_is_refused(dev):
    Correlation, _, _ = _mods()
    t1 = torch.randn(1, 4, 20, 24, device=dev, requires_grad=True)
    out = Correlation(8, 1, 8, 2, 2, 1)(t1, torch.randn(1, 4, 20, 24, device=dev))
    with pytest.raises(NotImplementedError):
        out.sum().backward()


@pytest.mark.parametrize("N,C,H,W,scale", [(2, 3, 33, 47, 4.0), (1, 2, 64, 64, 40.0), (1, 3, 1, 5, 2.0), (3, 1, 8, 8, 0.0)])
def test_resample_vs_oracle(dev, N, C, H, W, scale):
    _, Resample2d, _ = _mods()
    rng = np.random.default_rng(H * W)
    img = rng.standard_normal((N, C, H, W)).astype(np.float32)
    flow = (rng.standard_normal((N, 2, H, W)) * scale).astype(np.float32)
    ti, tf = cu(img, dev, True), cu(flow, dev, True)
    out = Resample2d()(ti, tf)
    close(out, O.resample2d_fwd(img, flow))
    go = rng.standard_normal(img.shape).astype(np.float32)
    out.backward(cu(go, dev))
    gi, gf = O.resample2d_bwd(img, flow, go)
    close(ti.grad, gi, atol=5e-5)  # float atomics: order-dependent last bits
    close(tf.grad, gf, atol=5e-5)


def test_resample_rejects_noncontiguous_flow_and_kernel_size(dev):
    _, Resample2d, _ = _mods()
    img = torch.randn(1, 3, 8, 8, device=dev)
    flow = torch.randn(1, 8, 8, 2, device=dev).permute(0, 3, 1, 2)
    with pytest.raises(ValueError):
        Resample2d()(img, flow)
    with pytest.raises(NotImplementedError):
        Resample2d(kernel_size=2)(img, flow.contiguous())


@pytest.mark.parametrize("N,C,H,W", [(2, 3, 32, 48), (1, 2, 7, 9), (1, 1, 1, 1), (2, 5, 3, 4)])
def test_channelnorm_vs_oracle(dev, N, C, H, W):
    _, _, ChannelNorm = _mods()
    rng = np.random.default_rng(C + H)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    x[0, :, 0, 0] = 0
    tx = cu(x, dev, True)
    out = ChannelNorm()(tx)
    ref = O.channelnorm_fwd(x)
    close(out, ref, atol=1e-6)
    go = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(cu(go, dev))
    close(tx.grad, O.channelnorm_bwd(x, ref, go))


def test_empty_batch(dev):
    Correlation, Resample2d, ChannelNorm = _mods()
    assert ChannelNorm()(torch.zeros(0, 3, 4, 4, device=dev)).shape == (0, 1, 4, 4)
    assert Resample2d()(torch.zeros(0, 3, 4, 4, device=dev), torch.zeros(0, 2, 4, 4, device=dev)).shape == (0, 3, 4, 4)
    assert Correlation(20, 1, 20, 1, 2, 1)(torch.zeros(0, 8, 4, 8, device=dev), torch.zeros(0, 8, 4, 8, device=dev)).shape == (0, 441, 4, 8)


def test_fused_warp_diff_norm(dev):
    from ir2rgb_amd.ext import warp_diff_norm
    rng = np.random.default_rng(11)
    im1 = rng.standard_normal((2, 3, 24, 40)).astype(np.float32)
    im2 = rng.standard_normal((2, 3, 24, 40)).astype(np.float32)
    flow = (rng.standard_normal((2, 2, 24, 40)) * 5).astype(np.float32)
    w, d, n = warp_diff_norm(cu(im1, dev), cu(im2, dev), cu(flow, dev))
    rw = O.resample2d_fwd(im2, flow)
    close(w, rw)
    close(d, im1 - rw)
    close(n, O.channelnorm_fwd(im1 - rw))
    w2, d2, n2 = warp_diff_norm(cu(im1, dev), cu(im2, dev), cu(flow, dev), want_warped=False, want_diff=False)
    assert w2 is None and d2 is None
    close(n2, O.channelnorm_fwd(im1 - rw))


# ---------------------------------------------------------------- full BASELINE sizes: properties
def test_correlation_full_size_properties(dev):
    """[1,256,64,128] (config 3): bilinearity, swap symmetry, centre channel = mean(f1*f2),
    and a spot check of random output rows against the oracle restricted to those rows."""
    Correlation, _, _ = _mods()
    corr = Correlation(20, 1, 20, 1, 2, 1)
    g = torch.Generator(device="cpu").manual_seed(0)
    f1 = torch.randn(1, 256, 64, 128, generator=g).to(dev)
    f2 = torch.randn(1, 256, 64, 128, generator=g).to(dev)
    f3 = torch.randn(1, 256, 64, 128, generator=g).to(dev)
    out = corr(f1, f2)
    assert out.shape == (1, 441, 64, 128)
    # centre displacement (tj=ti=0 -> channel 220) is the plain channel mean of the product
    torch.testing.assert_close(out[:, 220], (f1 * f2).mean(1), atol=1e-5, rtol=0)
    # linear in the second argument
    torch.testing.assert_close(corr(f1, f2 + 0.5 * f3), out + 0.5 * corr(f1, f3), atol=2e-5, rtol=0)
    # swapping the inputs mirrors the displacement: out12[tj,ti](y,x) = out21[-tj,-ti](y+2tj, x+2ti)
    out21 = corr(f2, f1).view(1, 21, 21, 64, 128)
    o12 = out.view(1, 21, 21, 64, 128)
    for tj, ti in [(3, -4), (-10, 10), (0, 7)]:
        a = o12[0, tj + 10, ti + 10]
        b = out21[0, -tj + 10, -ti + 10]
        ys = slice(max(0, -2 * tj), min(64, 64 - 2 * tj))
        xs = slice(max(0, -2 * ti), min(128, 128 - 2 * ti))
        ys2 = slice(ys.start + 2 * tj, ys.stop + 2 * tj)
        xs2 = slice(xs.start + 2 * ti, xs.stop + 2 * ti)
        torch.testing.assert_close(a[ys, xs], b[ys2, xs2], atol=1e-5, rtol=0)
    # rows 0..7 depend only on f2 rows 0..27: oracle on the cropped problem pins absolute values
    ref = O.correlation_fwd(f1[:, :, :8].cpu().numpy(), f2[:, :, :8].cpu().numpy(), 20, 1, 20, 1, 2)
    # (tj <= 0 reads f2 rows <= y only, so the crop does not change those displacement rows)
    close(out.view(1, 21, 21, 64, 128)[:, :11, :, :8], ref.reshape(1, 21, 21, 8, 128)[:, :11])


def test_resample_channelnorm_full_size_properties(dev):
    """512x1024 (config 3): identity for zero flow, integer shift == roll away from the border,
    out-of-range flow == border value; channelnorm homogeneity."""
    _, Resample2d, ChannelNorm = _mods()
    g = torch.Generator(device="cpu").manual_seed(1)
    img = torch.randn(1, 3, 512, 1024, generator=g).to(dev)
    zero = torch.zeros(1, 2, 512, 1024, device=dev)
    assert torch.equal(Resample2d()(img, zero), img)
    shift = zero.clone()
    shift[:, 0] = 3.0
    shift[:, 1] = -2.0
    out = Resample2d()(img, shift)
    assert torch.equal(out[:, :, 2:, :-3], img[:, :, :-2, 3:])
    far = zero.clone()
    far[:, 0] = 5000.0
    assert torch.equal(Resample2d()(img, far), img[:, :, :, -1:].expand_as(img))
    n = ChannelNorm()(img)
    torch.testing.assert_close(ChannelNorm()(img * -3.0), n * 3.0, atol=1e-5, rtol=1e-6)
    torch.testing.assert_close(n, img.pow(2).sum(1, keepdim=True).sqrt(), atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,C,H,W,lda,offa", [(1, 256, 8, 128, 256, 0), (2, 128, 9, 38, 192, 64), (1, 256, 5, 21, 256, 0),
                                                (1, 256, 24, 96, 320, 64)])
def test_correlation_nhwc_half_vs_oracle(dev, dtype, N, C, H, W, lda, offa):
    """ir2rgb_correlation_nhwc_half (banded MFMA products on half NHWC feature maps, the form FlowNetC's cost volume takes in
    this package's FlowNet2) against the C oracle of the reference operator on the same half-rounded inputs: products are
    exact, only the order of the fp32 sums differs.  Both output forms: fp32 planes (the operator's layout) and the half
    NHWC channel slice with LeakyReLU(0.1) that conv3_1 reads.  Odd widths, channel-slice views, rows and columns whose
    displaced partner lies outside the image."""
    import ctypes
    from ir2rgb_amd import _lib, conv as CV
    rng = np.random.default_rng(C + H + W)
    fa = torch.from_numpy(rng.standard_normal((N, C, H, W)).astype(np.float32)).to(dev).to(dtype)
    fb = torch.from_numpy(rng.standard_normal((N, C, H, W)).astype(np.float32)).to(dev).to(dtype)
    bufa = torch.zeros((N, lda, H, W), dtype=dtype, device=dev).contiguous(memory_format=torch.channels_last)
    bufb = torch.zeros((N, C, H, W), dtype=dtype, device=dev).contiguous(memory_format=torch.channels_last)
    bufa[:, offa:offa + C] = fa
    bufb[:] = fb
    ref = O.correlation_fwd(fa.float().cpu().numpy(), fb.float().cpu().numpy(), 20, 1, 20, 1, 2)
    lib = _lib.lib()
    dt = CV._TORCH2DT[dtype]
    out = torch.full((N, 441, H, W), float("nan"), dtype=torch.float32, device=dev)
    rc = lib.ir2rgb_correlation_nhwc_half(CV._p(bufa), lda, offa, CV._p(bufb), C, 0, CV._p(out), 0, 0, 0, 1.0, N, C, H, W, dt,
                                          _lib.current_stream(out))
    assert rc == 0
    close(out, ref, atol=2e-6 * np.sqrt(C))
    # half NHWC slice + LeakyReLU(0.1) into channels [32, 473) of a 512-channel buffer; the other channels stay untouched
    merged = torch.full((N, 512, H, W), 7.0, dtype=dtype, device=dev).contiguous(memory_format=torch.channels_last)
    rc = lib.ir2rgb_correlation_nhwc_half(CV._p(bufa), lda, offa, CV._p(bufb), C, 0, CV._p(merged), 1, 512, 32, 0.1, N, C, H, W, dt,
                                          _lib.current_stream(out))
    assert rc == 0
    want = torch.nn.functional.leaky_relu(torch.from_numpy(ref).to(dev), 0.1).to(dtype)
    got = merged[:, 32:473]
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert ((got.float() - want.float()).abs() <= ulp * want.float().abs() + 1e-6).all()
    assert (merged[:, :32] == 7.0).all() and (merged[:, 473:] == 7.0).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N", [1, 3])
def test_correlation_nhwc_half_full_size(dev, dtype, N):
    """ir2rgb_correlation_nhwc_half at the sizes the training window runs it (FlowNetC's conv3 outputs of a 512x1024 frame
    pair: [N,256,64,128], N = pairs per window = 1..3; flownet2_hip.py, reference FlowNetC.py:31 + correlation_cuda_kernel.cu:73-147).
    Absolute values: the C oracle on the half-rounded inputs restricted to the top and the bottom eight rows (displacement
    rows tj <= 0 read f2 rows <= y only, tj >= 0 rows >= y only, so a crop leaves those planes unchanged), every sample;
    everything in between: the fp32 operator of this package (itself pinned by the oracle) on the same inputs."""
    from ir2rgb_amd import _lib, conv as CV
    Correlation, _, _ = _mods()
    C, H, W = 256, 64, 128
    g = torch.Generator(device="cpu").manual_seed(10 + N)
    fa = torch.randn(N, C, H, W, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    fb = torch.randn(N, C, H, W, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    out = torch.full((N, 441, H, W), float("nan"), dtype=torch.float32, device=dev)
    rc = _lib.lib().ir2rgb_correlation_nhwc_half(CV._p(fa), C, 0, CV._p(fb), C, 0, CV._p(out), 0, 0, 0, 1.0, N, C, H, W,
                                                 CV._TORCH2DT[dtype], _lib.current_stream(out))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    tol = 2e-6 * np.sqrt(C)
    a32, b32 = fa.float().contiguous(), fb.float().contiguous()
    o5 = out.view(N, 21, 21, H, W)
    top = O.correlation_fwd(a32[:, :, :8].cpu().numpy(), b32[:, :, :8].cpu().numpy(), 20, 1, 20, 1, 2).reshape(N, 21, 21, 8, W)
    close(o5[:, :11, :, :8], top[:, :11], atol=tol)
    bot = O.correlation_fwd(a32[:, :, -8:].cpu().numpy(), b32[:, :, -8:].cpu().numpy(), 20, 1, 20, 1, 2).reshape(N, 21, 21, 8, W)
    close(o5[:, 10:, :, -8:], bot[:, 10:], atol=tol)
    ref = Correlation(20, 1, 20, 1, 2, 1)(a32, b32)
    torch.testing.assert_close(out, ref, atol=2 * tol, rtol=0)
    # the pipeline's form: half NHWC slice + LeakyReLU(0.1) inside a wider buffer
    merged = torch.full((N, 512, H, W), 7.0, dtype=dtype, device=dev).contiguous(memory_format=torch.channels_last)
    rc = _lib.lib().ir2rgb_correlation_nhwc_half(CV._p(fa), C, 0, CV._p(fb), C, 0, CV._p(merged), 1, 512, 32, 0.1, N, C, H, W,
                                                 CV._TORCH2DT[dtype], _lib.current_stream(out))
    assert rc == 0
    want = torch.nn.functional.leaky_relu(out, 0.1).to(dtype)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert ((merged[:, 32:473].float() - want.float()).abs() <= ulp * want.float().abs() + 1e-6).all()
    assert (merged[:, :32] == 7.0).all() and (merged[:, 473:] == 7.0).all()
