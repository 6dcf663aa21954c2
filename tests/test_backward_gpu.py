"""GPU gradient parity: loss.backward() through the HIP stages vs torch autograd through the
plain-torch fp32 restatement (oracle/networks_oracle.py, itself pinned to the reference goldens).

Both sides start from the same seeded parameters and inputs.  Exactness of every stage's backward is
established one stage deep in tests/test_stage_backward_gpu.py (<= 6e-3 f16 / 3e-2 bf16 against fp32
autograd).  Here whole networks are chained: 12..40 half-precision layers whose BatchNorm statistics
come from as few as 32 pixels and whose ReLU masks flip under rounding, so the bound is a noise
bound, per parameter tensor, relative L2 (measured worst cases in parentheses):
    local generator   f16 <= 1e-1 (6.1e-2)    bf16 <= 2.5e-1 (1.5e-1)
    discriminator     f16 <= 6e-2 (3.4e-2)    bf16 <= 2e-1   (1.0e-1)
    coarse generator  f16 <= 2.5e-1 (1.2e-1)
Tensors whose oracle gradient is numerically zero (conv biases in front of BatchNorm) are skipped.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

OPT = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)


def _rel(a, b):
    return ((a.float().cpu() - b).norm() / b.norm().clamp_min(1e-20)).item()


def _smooth(shape, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (2, 2, 2, 2), mode="reflect"), 5, stride=1)
    return torch.tanh(x * 3)


def _tame_flow(g):
    """Random-init flow heads emit +-20..40 px flows; the warp's gradient w.r.t. the flow is piecewise
    constant per pixel cell, so at that magnitude half-precision rounding of the flow flips cells and
    the comparison measures chaos, not kernels.  Scale the head down to sub-pixel flows (what a trained
    network produces on slow motion) in BOTH modules."""
    with torch.no_grad():
        g.model_final_flow[1].weight.mul_(0.01)
        g.model_final_flow[1].bias.mul_(0.01)


def _compare_grads(hip_mod, ref_mod, tol, skip_tiny=1e-7):
    bad, worst = {}, 0.0
    ref_norms = {k: p.grad.norm().item() for k, p in ref_mod.named_parameters() if p.grad is not None}
    scale = max(ref_norms.values())
    for (k, p), (_, q) in zip(hip_mod.named_parameters(), ref_mod.named_parameters()):
        assert q.grad is not None, k
        if q.grad.norm().item() < skip_tiny * scale:
            continue  # analytically zero (bias before BatchNorm)
        assert p.grad is not None, f"no gradient for {k}"
        e = _rel(p.grad, q.grad)
        worst = max(worst, e)
        if not e <= tol:
            bad[k] = e
    assert not bad, f"gradient relative L2 over {tol}: {dict(sorted(bad.items(), key=lambda kv: -kv[1])[:10])} ... worst {worst}, n_bad {len(bad)}"
    return worst


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-1), (torch.bfloat16, 2.5e-1)])
def test_local_generator_gradients(dev, dtype, tol):
    from ir2rgb_amd import networks as N
    from oracle import networks_oracle as NO
    torch.manual_seed(21)
    g = N.build_generator_module(9, 3, 6, 64, "composite-local", 3, "batch", 1, **OPT).train()
    _tame_flow(g)
    ref = copy.deepcopy(g)
    A, P = _smooth((1, 9, 32, 48), 1), _smooth((1, 6, 32, 48), 2)
    fi, ff = _smooth((1, 128, 16, 24), 3).abs(), _smooth((1, 128, 16, 24), 4).abs()
    proj = [_smooth((1, c, 32, 48), 10 + c) for c in (3, 2, 1, 3)]

    def loss_of(outs):
        final, flow, weight, raw = outs[0], outs[1], outs[2], outs[3]
        return ((final * proj[0].to(final.device)).sum() + (flow * proj[1].to(final.device)).sum() * 0.05 +
                (weight * proj[2].to(final.device)).sum() + (raw * proj[3].to(final.device)).sum())

    fir = fi.clone().requires_grad_()
    loss_ref = loss_of(NO.generator_forward(ref, A, P, fir, ff))
    loss_ref.backward()

    g = g.to(dev)
    g.compute_dtype = dtype
    fih = fi.to(dev).requires_grad_()
    loss_hip = loss_of(g(A.to(dev), P.to(dev), None, fih, ff.to(dev), None, False))
    loss_hip.backward()
    assert abs(loss_hip.item() - loss_ref.item()) <= tol * abs(loss_ref.item()) + 1.0
    worst = _compare_grads(g, ref, tol)
    e_in = _rel(fih.grad, fir.grad)
    print(f"local generator {dtype}: worst param-grad rel L2 {worst:.4f}, coarse-feature grad {e_in:.4f}")
    assert e_in <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 6e-2), (torch.bfloat16, 2e-1)])
def test_discriminator_gradients(dev, dtype, tol):
    from ir2rgb_amd import networks as N
    from oracle import networks_oracle as NO
    torch.manual_seed(22)
    d = N.build_discriminator_module(6, 64, 3, "batch", 2, True).train()
    ref = copy.deepcopy(d)
    x = _smooth((2, 6, 48, 64), 5)

    def loss_of(outs):
        # LSGAN-style on the logits plus a feature term on every intermediate (as GAN_and_FM_loss does)
        total = 0
        for sc in outs:
            total = total + ((sc[-1].float() - 1) ** 2).mean()
            for f in sc[:-1]:
                total = total + f.float().abs().mean() * 0.1
        return total

    xr = x.clone().requires_grad_()
    loss_of(NO.discriminator_forward(ref, xr)).backward()
    d = d.to(dev)
    d.compute_dtype = dtype
    xh = x.to(dev).requires_grad_()
    loss_of(d(xh)).backward()
    worst = _compare_grads(d, ref, tol)
    e_in = _rel(xh.grad, xr.grad)
    print(f"discriminator {dtype}: worst param-grad rel L2 {worst:.4f}, input grad {e_in:.4f}")
    assert e_in <= tol


def test_composite_generator_gradients_f16(dev):
    """Coarse generator (encoders, 9 ResnetBlocks, transposed convolutions) at ngf=64."""
    from ir2rgb_amd import networks as N
    from oracle import networks_oracle as NO
    torch.manual_seed(23)
    g = N.build_generator_module(9, 3, 6, 64, "composite", 3, "batch", 0, **OPT).train()
    _tame_flow(g)
    ref = copy.deepcopy(g)
    A, P = _smooth((1, 9, 32, 64), 1), _smooth((1, 6, 32, 64), 2)
    pr = _smooth((1, 3, 32, 64), 7)

    def loss_of(outs):
        return (outs[0] * pr.to(outs[0].device)).sum() + (outs[1] ** 2).mean() * 0.01 + outs[2].sum() * 0.01

    loss_of(NO.generator_forward(ref, A, P)).backward()
    g = g.to(dev)
    g.compute_dtype = torch.float16
    loss_of(g(A.to(dev), P.to(dev), None, None, None, None, False)).backward()
    worst = _compare_grads(g, ref, 2.5e-1)
    print(f"composite generator f16: worst param-grad rel L2 {worst:.4f}")
