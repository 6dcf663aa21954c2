"""CPU tests: the C oracle vs the committed golden vectors and vs the closed forms."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import closed_form as cf
from oracle import ops

T = torch.from_numpy


def _load(golden_dir, pattern):
    files = sorted(glob.glob(os.path.join(golden_dir, pattern)))
    assert files, pattern
    return [(os.path.basename(f), np.load(f)) for f in files]


def test_correlation_golden(golden_dir):
    for name, g in _load(golden_dir, "ops_corr_*.npz"):
        hp = [int(v) for v in g["hp"]]
        out = ops.correlation_fwd(g["f1"], g["f2"], *hp)
        np.testing.assert_allclose(out, g["out"], rtol=0, atol=1e-7, err_msg=name)
        g1, g2 = ops.correlation_bwd(g["f1"], g["f2"], g["gout"], *hp)
        np.testing.assert_allclose(g1, g["g1"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(g2, g["g2"], rtol=0, atol=1e-7)


def test_resample_golden(golden_dir):
    for name, g in _load(golden_dir, "ops_resample_*.npz"):
        np.testing.assert_array_equal(ops.resample2d_fwd(g["img"], g["flow"]), g["out"], err_msg=name)
        gi, gf = ops.resample2d_bwd(g["img"], g["flow"], g["gout"])
        np.testing.assert_array_equal(gi, g["gimg"])
        np.testing.assert_array_equal(gf, g["gflow"])


def test_channelnorm_golden(golden_dir):
    for name, g in _load(golden_dir, "ops_cnorm_*.npz"):
        np.testing.assert_array_equal(ops.channelnorm_fwd(g["x"]), g["out"], err_msg=name)
        np.testing.assert_array_equal(ops.channelnorm_bwd(g["x"], g["out"], g["gout"]), g["gin"])


@pytest.mark.parametrize("C,H,W", [(3, 5, 9), (16, 12, 20)])
def test_correlation_closed_form(C, H, W):
    rng = np.random.default_rng(C * 100 + H)
    f1 = rng.standard_normal((2, C, H, W)).astype(np.float32)
    f2 = rng.standard_normal((2, C, H, W)).astype(np.float32)
    hp = (20, 1, 20, 1, 2)
    out = ops.correlation_fwd(f1, f2, *hp)
    assert out.shape == (2, 441, H, W)
    np.testing.assert_allclose(out, cf.correlation(T(f1), T(f2), *hp).numpy(), atol=2e-6)
    # autograd of the closed form pins the backward restatement
    go = rng.standard_normal(out.shape).astype(np.float32)
    t1, t2 = T(f1).requires_grad_(), T(f2).requires_grad_()
    cf.correlation(t1, t2, *hp).backward(T(go))
    g1, g2 = ops.correlation_bwd(f1, f2, go, *hp)
    np.testing.assert_allclose(g1, t1.grad.numpy(), atol=2e-6)
    np.testing.assert_allclose(g2, t2.grad.numpy(), atol=2e-6)


def test_correlation_out_shape_other_params():
    # kernel_size 3, stride1 2, displacement not divisible by stride2 (20 // 3 = 6 -> 13x13)
    assert ops.correlation_out_shape(8, 24, 32, 21, 3, 20, 2, 3) == (169, 12, 16)
    assert ops.correlation_out_shape(256, 64, 128, 20, 1, 20, 1, 2) == (441, 64, 128)


def test_resample_closed_form_and_identity():
    rng = np.random.default_rng(3)
    img = rng.standard_normal((1, 3, 10, 14)).astype(np.float32)
    zero = np.zeros((1, 2, 10, 14), np.float32)
    np.testing.assert_array_equal(ops.resample2d_fwd(img, zero), img)  # zero flow = identity
    flow = (rng.standard_normal((1, 2, 10, 14)) * 30).astype(np.float32)  # mostly out of range
    np.testing.assert_allclose(ops.resample2d_fwd(img, flow), cf.resample2d(T(img), T(flow)).numpy(), atol=2e-5)


def test_resample_backward_matches_autograd_for_positive_coords():
    """Where xf,yf >= 0 truncation == floor, so the reference's image gradient equals the
    autograd gradient of the closed form; the flow gradient equals it wherever no corner is
    clamped."""
    rng = np.random.default_rng(5)
    H, W = 12, 16
    img = rng.standard_normal((1, 3, H, W)).astype(np.float32)
    flow = rng.uniform(0.05, 0.95, (1, 2, H, W)).astype(np.float32)
    flow[:, :, -1, :] = 0.25
    flow[:, :, :, -1] = 0.25
    go = rng.standard_normal((1, 3, H, W)).astype(np.float32)
    ti, tf = T(img).requires_grad_(), T(flow).requires_grad_()
    cf.resample2d(ti, tf).backward(T(go))
    gi, gf = ops.resample2d_bwd(img, flow, go)
    np.testing.assert_allclose(gi, ti.grad.numpy(), atol=1e-5)
    np.testing.assert_allclose(gf[:, :, :-1, :-1], tf.grad.numpy()[:, :, :-1, :-1], atol=1e-4)


def test_channelnorm_closed_form_and_grad():
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 3, 6, 10)).astype(np.float32)
    out = ops.channelnorm_fwd(x)
    np.testing.assert_allclose(out, cf.channelnorm(T(x)).numpy(), atol=1e-6)
    go = rng.standard_normal(out.shape).astype(np.float32)
    tx = T(x).requires_grad_()
    cf.channelnorm(tx).backward(T(go))
    np.testing.assert_allclose(ops.channelnorm_bwd(x, out, go), tx.grad.numpy(), atol=1e-5)


def test_empty_inputs():
    assert ops.channelnorm_fwd(np.zeros((0, 3, 4, 4), np.float32)).shape == (0, 1, 4, 4)
    assert ops.resample2d_fwd(np.zeros((0, 3, 4, 4), np.float32), np.zeros((0, 2, 4, 4), np.float32)).shape == (0, 3, 4, 4)
