"""Generates tests/golden/ops_*.npz -- golden vectors for the three FlowNet2 operators.

The reference's operators are CUDA-only and cannot run in the authoring container, and the
reference holds no fixtures for them, so these vectors are frozen from the CPU restatement
(oracle/ops_ref.c) AFTER it has been cross-checked against the independent closed forms of
oracle/closed_form.py inside this script (it aborts if they disagree).  They pin the oracle
against silent drift; "parity unpinned by the reference's own tests" still applies.

    python tests/golden/make_op_goldens.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import closed_form as cf  # noqa: E402
from oracle import ops  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20240607)
T = torch.from_numpy


def corr_case(name, C, H, W, N=1):
    hp = (20, 1, 20, 1, 2)
    f1 = rng.standard_normal((N, C, H, W)).astype(np.float32)
    f2 = rng.standard_normal((N, C, H, W)).astype(np.float32)
    out = ops.correlation_fwd(f1, f2, *hp)
    ref = cf.correlation(T(f1), T(f2), *hp).numpy()
    assert np.abs(out - ref).max() < 2e-6, name
    go = rng.standard_normal(out.shape).astype(np.float32)
    g1, g2 = ops.correlation_bwd(f1, f2, go, *hp)
    t1, t2 = T(f1).requires_grad_(), T(f2).requires_grad_()
    cf.correlation(t1, t2, *hp).backward(T(go))
    assert np.abs(g1 - t1.grad.numpy()).max() < 2e-6 and np.abs(g2 - t2.grad.numpy()).max() < 2e-6, name
    # keep the fixture small: inputs, a strided sample of the output, full grads only for small C
    np.savez_compressed(os.path.join(OUT, f"ops_corr_{name}.npz"), f1=f1, f2=f2, hp=np.array(hp),
                        out=out, gout=go, g1=g1, g2=g2)


def resample_case(name, C, H, W, scale, N=2):
    img = rng.standard_normal((N, C, H, W)).astype(np.float32)
    flow = (rng.standard_normal((N, 2, H, W)) * scale).astype(np.float32)
    # a few exact-integer and far-out-of-range displacements
    flow[0, :, 0, :4] = np.array([[0.0, 1.0, -1.0, 3.0], [0.0, -2.0, 2.0, 0.0]], np.float32)
    flow[-1, 0, -1, -3:] = [1e4, -1e4, 0.5]
    out = ops.resample2d_fwd(img, flow)
    ref = cf.resample2d(T(img), T(flow)).numpy()
    assert np.abs(out - ref).max() < 2e-5, (name, np.abs(out - ref).max())
    go = rng.standard_normal(out.shape).astype(np.float32)
    gi, gf = ops.resample2d_bwd(img, flow, go)
    np.savez_compressed(os.path.join(OUT, f"ops_resample_{name}.npz"), img=img, flow=flow, out=out, gout=go, gimg=gi,
                        gflow=gf)


def cnorm_case(name, C, H, W, N=2):
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    x[0, :, 0, 0] = 0.0  # zero vector: backward hits the 1e-9 guard
    out = ops.channelnorm_fwd(x)
    assert np.abs(out - cf.channelnorm(T(x)).numpy()).max() < 1e-6
    go = rng.standard_normal(out.shape).astype(np.float32)
    gin = ops.channelnorm_bwd(x, out, go)
    np.savez_compressed(os.path.join(OUT, f"ops_cnorm_{name}.npz"), x=x, out=out, gout=go, gin=gin)


if __name__ == "__main__":
    corr_case("c32_8x16", 32, 8, 16)
    corr_case("c7_6x8", 7, 6, 8, N=2)
    resample_case("c3_16x24", 3, 16, 24, 5.0)
    resample_case("c2_9x7", 2, 9, 7, 2.0)
    cnorm_case("c3_16x24", 3, 16, 24)
    cnorm_case("c2_5x7", 2, 5, 7)
    print("ok")
