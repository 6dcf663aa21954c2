"""Whole-network forward + backward parity of the HIP stages against the ROUNDING-EMULATING oracle
(oracle/emulated.py: the reference's graph, fp32 torch operators and torch.autograd derivatives, values rounded to the
compute dtype exactly where the HIP path stores a half tensor; pinned to the reference goldens by
tests/test_oracle_networks.py).

What such a comparison can and cannot show (measured, see the printed values):
* Two evaluations that round at the same places still differ in fp32 summation order (1e-6).  That flips the rounding of
  ~5e-4 of a layer's outputs by one ulp; each flipped input moves every output of the next layer by ~1e-4 of its spread,
  which flips ~3 % of THOSE roundings -- after 3-4 layers every element carries an independent one-ulp error, as between
  any two half-precision evaluations.  The discriminators (5 layers) show the onset: 0 / 5e-5 / 3e-4 / 1e-3 / 1.3e-3 per
  layer (bf16).  Past ~5 layers the emulation is no closer to the HIP path than the fp32 oracle is.
* So the per-tensor relative L2 of a gradient is a NOISE measurement (a few 1e-2 for 5 layers, 5e-2 .. 2.7e-1 for 12 .. 40
  layers with BatchNorm over 32 .. 128 pixels), and a bound on it cannot catch a missing 10 % term in a deep network.
* Rounding noise is zero-mean and lives in a space of 1e3 .. 1e6 dimensions; a structural error (a dropped residual path,
  a mis-scaled BatchNorm term, a wrong tap) is not.  The statistic that separates them is the PROJECTION of the HIP
  gradient on the oracle's, <g, r> / <r, r>: noise of relative size e moves it by ~e^2 at most (measured <= 2e-2 even
  where the relative L2 is 2.7e-1); scaling any stage's contribution by 0.9 moves it by up to 1e-1.

Bounds, per tensor:
    forward outputs, relative L2          5 layers 5e-3 | <= 12 layers 5e-3 (f16) 1.5e-2 (bf16) | 40 layers 5e-3 / 2e-2
    gradients, relative L2 (noise)        5 layers 3e-2 | <= 12 layers 8e-2 / 2e-1              | 40 layers 1.5e-1 / 3.5e-1
    gradients, |projection - 1|           3e-2 (f16) / 6e-2 (bf16) for every tensor of >= 512 elements
Every stage is also checked one stage deep at 1e-2 / 4e-2 against fp32 autograd (tests/test_stage_backward_gpu.py), and
losses, outputs and gradients of whole training windows against reference-generated goldens (tests/test_harness_gpu.py).

The flow head is scaled to emit flows of a few pixels (x0.05 on both sides): with random-init weights it emits +-40 px,
and the warp's derivative w.r.t. the flow is piecewise constant per pixel cell, so at that magnitude a 1e-3 relative
difference in the flow flips cells and the comparison would measure that, not the kernels.
"""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

OPT = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)
FWD_TOL, GRAD_TOL = 5e-3, 3e-2


def _rel(a, b):
    b = b.detach().float().cpu()
    return ((a.detach().float().cpu() - b).norm() / b.norm().clamp_min(1e-20)).item()


def _smooth(shape, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (2, 2, 2, 2), mode="reflect"), 5, stride=1)
    return torch.tanh(x * 3)


def _tame_flow(g):
    with torch.no_grad():
        g.model_final_flow[1].weight.mul_(0.05)
        g.model_final_flow[1].bias.mul_(0.05)


PROJ_TOL = {torch.float16: 3e-2, torch.bfloat16: 6e-2}


def _proj(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return (a @ b / (b @ b)).item()


def _compare_grads(hip_mod, ref_mod, tol, what, proj_tol):
    ref_norms = {k: p.grad.norm().item() for k, p in ref_mod.named_parameters() if p.grad is not None}
    scale = max(ref_norms.values())
    errs, projs = {}, {}
    for (k, p), (_, q) in zip(hip_mod.named_parameters(), ref_mod.named_parameters()):
        if q.grad is None or q.grad.norm().item() < 1e-6 * scale:
            # the bias of a convolution in front of BatchNorm: exactly zero on the HIP side (None or zeros)
            assert p.grad is None or p.grad.norm().item() <= 1e-4 * scale, k
            continue
        assert p.grad is not None, f"no gradient for {k}"
        errs[k] = _rel(p.grad, q.grad)
        if q.grad.numel() >= 512:
            projs[k] = abs(_proj(p.grad, q.grad) - 1.0)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    worst_p = sorted(projs.items(), key=lambda kv: -kv[1])[:4]
    print(what, "parameter gradients: worst relative L2", [(k, round(v, 4)) for k, v in worst], "worst |projection-1|",
          [(k, round(v, 4)) for k, v in worst_p], "of", len(errs))
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, f"{what}: gradient relative L2 over {tol}: {dict(sorted(bad.items(), key=lambda kv: -kv[1])[:10])}"
    # the flow head's gradient passes through the warp, piecewise constant in the flow: cell flips move it more
    badp = {k: v for k, v in projs.items() if not v <= (proj_tol if "final_flow" not in k else max(0.1, proj_tol))}
    assert not badp, f"{what}: gradient projection off by more than {proj_tol}: {dict(sorted(badp.items(), key=lambda kv: -kv[1])[:10])}"
    return max(errs.values())


DEEP_FWD_TOL = {torch.float16: 5e-3, torch.bfloat16: 2e-2}
DEEP_GRAD_TOL = {torch.float16: 1.5e-1, torch.bfloat16: 3.5e-1}
MID_FWD_TOL = {torch.float16: 5e-3, torch.bfloat16: 1.5e-2}
MID_GRAD_TOL = {torch.float16: 8e-2, torch.bfloat16: 2e-1}


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("depth", ["shallow", "full"])
@pytest.mark.parametrize("model", ["composite", "composite-local"])
def test_generator_forward_backward_vs_emulated(dev, model, depth, dtype):
    from ir2rgb_amd import networks as N
    from oracle import emulated as E
    local = model == "composite-local"
    torch.manual_seed(21 if local else 23)
    shallow = depth == "shallow"
    opt = dict(OPT, gen_blocks=2, n_blocks_local=1) if shallow else OPT
    fwd_tol, grad_tol = (MID_FWD_TOL[dtype], MID_GRAD_TOL[dtype]) if shallow else (DEEP_FWD_TOL[dtype], DEEP_GRAD_TOL[dtype])
    g = N.build_generator_module(9, 3, 6, 64, model, 2 if shallow else 3, "batch", 1 if local else 0, **opt).train()
    _tame_flow(g)
    ref = copy.deepcopy(g)
    H, W = (32, 48) if local else (32, 64)
    A, P = _smooth((1, 9, H, W), 1), _smooth((1, 6, H, W), 2)
    fi = ff = None
    if local:
        fi, ff = _smooth((1, 128, H // 2, W // 2), 3).abs(), _smooth((1, 128, H // 2, W // 2), 4).abs()
    proj = [_smooth((1, c, H, W), 10 + c) for c in (3, 2, 1, 3)]

    def loss_of(outs):
        final, flow, weight, raw = outs[0], outs[1], outs[2], outs[3]
        dv = final.device
        return ((final * proj[0].to(dv)).sum() + (flow * proj[1].to(dv)).sum() * 0.05 + (weight * proj[2].to(dv)).sum() +
                (raw * proj[3].to(dv)).sum())

    fir = fi.clone().requires_grad_() if local else None
    out_ref = E.generator_forward(ref, A, P, fir, ff, dtype=dtype)
    loss_of(out_ref).backward()

    g = g.to(dev)
    g.compute_dtype = dtype
    fih = fi.to(dev).requires_grad_() if local else None
    out = g(A.to(dev), P.to(dev), None, fih, ff.to(dev) if local else None, None, False)
    loss_of(out).backward()
    ferr = {n: _rel(out[i], out_ref[i]) for i, n in ((0, "img_final"), (1, "flow"), (2, "weight"), (3, "img_raw"), (4, "img_feat"),
                                                     (5, "flow_feat"))}
    print(model, depth, dtype, "forward", {k: round(v, 5) for k, v in ferr.items()})
    assert all(v <= fwd_tol for v in ferr.values()), ferr
    _compare_grads(g, ref, grad_tol, f"{model} {depth} {dtype}", PROJ_TOL[dtype])
    if local:
        e_in, p_in = _rel(fih.grad, fir.grad), _proj(fih.grad, fir.grad)
        print(model, depth, dtype, "coarse-feature gradient", e_in, "projection", p_in)
        assert e_in <= grad_tol and abs(p_in - 1) <= PROJ_TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("input_nc", [6, 13])
def test_discriminator_forward_backward_vs_emulated(dev, input_nc, dtype):
    from ir2rgb_amd import networks as N
    from ir2rgb_amd.losses import fused_losses
    from oracle import emulated as E
    torch.manual_seed(22)
    d = N.build_discriminator_module(input_nc, 64, 3, "batch", 2, True).train()
    ref = copy.deepcopy(d)
    x, x2 = _smooth((2, input_nc, 48, 64), 5), _smooth((2, input_nc, 48, 64), 6)

    # LSGAN term on the logits + feature matching against a second (constant) input, as GAN_and_FM_loss composes them
    xr = x.clone().requires_grad_()
    fr, cr = E.discriminator_forward(ref, xr, dtype), E.discriminator_forward(ref, x2, dtype)
    loss_ref = sum(((sc[-1] - 1) ** 2).mean() for sc in fr)
    for i in range(2):
        for j in range(4):
            loss_ref = loss_ref + E.l1_half(fr[i][j], cr[i][j], 0.5 * 0.8 * 10.0, dtype)
    loss_ref.backward()

    d = d.to(dev)
    d.compute_dtype = dtype
    xh = x.to(dev).requires_grad_()
    fh, ch = d(xh), d(x2.to(dev))
    terms = [("mse", sc[-1], 1.0, 1.0, 0) for sc in fh]
    terms += [("l1", fh[i][j], ch[i][j], 0.5 * 0.8 * 10.0, 0) for i in range(2) for j in range(4)]
    loss = fused_losses(terms, 1, dtype)[0]
    loss.backward()
    ferr = {f"out{i}_{j}": _rel(fh[i][j], fr[i][j]) for i in range(2) for j in range(5)}
    print("discriminator", input_nc, dtype, "forward", {k: round(v, 5) for k, v in ferr.items()},
          "loss", loss.item(), loss_ref.item())
    assert all(v <= FWD_TOL for v in ferr.values()), ferr
    assert abs(loss.item() - loss_ref.item()) <= 2e-3 * abs(loss_ref.item())
    _compare_grads(d, ref, GRAD_TOL, f"discriminator {input_nc} {dtype}", 1e-2)
    e_in, p_in = _rel(xh.grad, xr.grad), _proj(xh.grad, xr.grad)
    print("discriminator", input_nc, dtype, "input gradient", e_in, "projection", p_in)
    assert e_in <= GRAD_TOL and abs(p_in - 1) <= 1e-2
