"""GPU parity of the grouped loss kernels (ir2rgb_amd.losses / csrc/losses.hip) against the torch
formulas the reference evaluates: criterionFeat = nn.L1Loss on discriminator features
(discriminator.py:199-210), criterionGAN = least-squares GANLoss (loss.py), criterionFlow = MaskedL1Loss
(loss.py).  The oracle here is the formula in fp32/fp64 on the same inputs: values to 1e-5 relative
(the only difference is the fp32 summation order), gradients bit-exact up to the half rounding of the
scalar g*weight/n (L1) or to 1e-6 relative (fp32 terms)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_grouped_l1_and_mse_match_torch(dtype):
    from ir2rgb_amd.losses import fused_losses
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    shapes = [(2, 64, 33, 65), (1, 128, 17, 33), (3, 512, 5, 9), (1, 64, 128, 256)]
    feats_a = [torch.randn(s, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_() for s in shapes]
    feats_b = [torch.randn(s, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last) for s in shapes]
    feats_b[0][:, :, :3] = feats_a[0].detach()[:, :, :3]          # exact ties: sign(0) = 0
    logits = [torch.randn(2, 1, 35, 67, generator=g).to(dev).requires_grad_(), torch.randn(2, 1, 19, 35, generator=g).to(dev).requires_grad_()]
    ws = [0.5, 1.25, 2.0, 0.125]
    terms = [("mse", lg, 1.0, 1.0, 0) for lg in logits] + [("l1", a, b, w, 1) for a, b, w in zip(feats_a, feats_b, ws)]
    out = fused_losses(terms, 2, dtype)
    coef = torch.tensor([0.7, 1.3], device=dev)
    (out * coef).sum().backward()
    got = [t.grad.clone() for t in logits + feats_a]
    for t in logits + feats_a:
        t.grad = None
    ref_gan = sum(((lg.double() - 1.0) ** 2).mean() for lg in logits)
    ref_fm = sum(w * (a.double() - b.double()).abs().mean() for a, b, w in zip(feats_a, feats_b, ws))
    (ref_gan * 0.7 + ref_fm * 1.3).backward()
    assert abs(out[0].item() - ref_gan.item()) <= 1e-5 * abs(ref_gan.item())
    assert abs(out[1].item() - ref_fm.item()) <= 1e-5 * abs(ref_fm.item())
    for t, gg in zip(logits, got[:2]):
        assert torch.allclose(gg, t.grad, rtol=1e-5, atol=1e-9)
    for t, b, gg, w in zip(feats_a, feats_b, got[2:], ws):
        assert gg.dtype == dtype and gg.stride() == t.stride()
        scalar = (coef[1].cpu() * (torch.tensor(w, dtype=torch.float32) / torch.tensor(float(t.numel()), dtype=torch.float32))).to(dtype)
        want = torch.sign(t.detach().float() - b.float()) * scalar.float().item()   # the half-rounded g*weight/n
        assert torch.equal(gg.float(), want), "L1 gradient must be sign(a-b) * half(g*weight/n) exactly"


def test_masked_l1_matches_torch():
    from ir2rgb_amd.losses import fused_losses
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    a1 = torch.randn(2, 2, 40, 72, generator=g).to(dev).requires_grad_()
    b1 = torch.randn(2, 2, 40, 72, generator=g).to(dev)
    a2 = torch.randn(2, 3, 40, 72, generator=g).to(dev).requires_grad_()
    a3 = torch.rand(2, 1, 40, 72, generator=g).to(dev).requires_grad_()
    mask = (torch.rand(2, 1, 40, 72, generator=g) < 0.7).float().to(dev)
    b2 = torch.randn(2, 3, 40, 72, generator=g).to(dev)
    out = fused_losses([("ml1", a1, b1, mask, 5.0, 0), ("ml1", a2, b2, mask, 10.0, 1), ("ml1", a3, None, mask, 1.0, 2)], 4)
    out.sum().backward()
    got = [t.grad.clone() for t in (a1, a2, a3)]
    for t in (a1, a2, a3):
        t.grad = None
    F = torch.nn.functional
    refs = [F.l1_loss(a1 * mask, b1 * mask) * 5.0, F.l1_loss(a2 * mask, b2 * mask) * 10.0, F.l1_loss(a3 * mask, torch.zeros_like(a3))]
    sum(refs).backward()
    for k in range(3):
        assert abs(out[k].item() - refs[k].item()) <= 2e-5 * abs(refs[k].item())
    assert out[3].item() == 0.0
    for t, gg in zip((a1, a2, a3), got):
        assert torch.allclose(gg, t.grad, rtol=1e-6, atol=1e-12)


def test_loss_group_rejects_bad_input():
    from ir2rgb_amd.losses import fused_losses
    dev = _dev()
    a = torch.randn(1, 8, 3, 3, device=dev).bfloat16()            # 72 elements: not a multiple of 8 -> fine; 9*8
    with pytest.raises(ValueError):
        fused_losses([("l1", a[:, :7], a[:, :7], 1.0, 0)], 1)      # 63 elements
    with pytest.raises(ValueError):
        fused_losses([("mse", a.float(), 1.0, 1.0, 2)], 2)         # slot out of range
    with pytest.raises(ValueError):
        fused_losses([("l1", a.cpu(), a.cpu(), 1.0, 0)], 1)        # no CPU fallback


def test_training_window_losses_fused_vs_torch_ops():
    """The same first training window through the grouped kernels and through torch ops: every reported
    loss agrees to 2e-3 relative (bf16 feature differences are rounded to bf16 by torch before the mean,
    the grouped kernel keeps them in fp32)."""
    from ir2rgb_amd import vid2vid as V
    dev = _dev()
    A, B = V.synthetic_sequence(4, 64, 128, 7, dev)
    res = []
    for fused in (True, False):
        tr = V.Vid2VidTrainer(dev, seed=0, first_layer_gen_filters=64, gen_blocks=2, fused_losses=fused)
        res.append({k: v.item() for k, v in tr.train_window(A[:, :3], B[:, :3]).items()})
    for k in res[0]:
        assert abs(res[0][k] - res[1][k]) <= 2e-3 * abs(res[1][k]), (k, res[0][k], res[1][k])


def test_fused_adam_param_groups_and_state_dict():
    """The optimizer surface the reference touches: ``param_groups[...]['lr']`` (Model.update_learning_rate,
    base_model.py:103-108), zero_grad, and a state_dict in torch.optim.Adam's layout -- a run can move between the two
    optimizers in either direction and continue with the same updates."""
    from ir2rgb_amd.optim import FusedAdam
    dev = _dev()
    g = torch.Generator().manual_seed(12)
    init = [torch.randn(s, generator=g) for s in [(5,), (16, 3, 3, 3), (257,)]]
    mine = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    ref = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    opt_m, opt_r = FusedAdam(mine, lr=2e-3, betas=(0.5, 0.999)), torch.optim.Adam(ref, lr=2e-3, betas=(0.5, 0.999))

    def step(opts, lists):
        grads = [torch.randn(p.shape, generator=g).to(dev) for p in lists[0]]
        for lst in lists:
            for p, gr in zip(lst, grads):
                p.grad = gr.clone()
        for o in opts:
            o.step()

    step([opt_m, opt_r], [mine, ref])
    for o in (opt_m, opt_r):                       # the reference's learning-rate decay
        for group in o.param_groups:
            group["lr"] = 1e-3
    step([opt_m, opt_r], [mine, ref])
    assert all(torch.allclose(p, q, rtol=2e-6, atol=1e-7) for p, q in zip(mine, ref))
    # FusedAdam -> torch.optim.Adam -> FusedAdam
    mine2 = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt_t = torch.optim.Adam(mine2, lr=1.0, betas=(0.9, 0.9))
    opt_t.load_state_dict(opt_m.state_dict())
    assert opt_t.param_groups[0]["lr"] == 1e-3 and tuple(opt_t.param_groups[0]["betas"]) == (0.5, 0.999)
    mine3 = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt_f = FusedAdam(mine3, lr=1.0, betas=(0.9, 0.9))
    opt_f.load_state_dict(opt_t.state_dict())
    assert opt_f.lr == 1e-3 and opt_f.step_count == 2
    step([opt_m, opt_t, opt_f, opt_r], [mine, mine2, mine3, ref])
    for a, b, c, d in zip(mine, mine2, mine3, ref):
        assert torch.allclose(a, d, rtol=2e-6, atol=1e-7) and torch.allclose(b, d, rtol=2e-6, atol=1e-7) and torch.equal(a, c)
    opt_m.zero_grad()
    assert all(p.grad is None for p in mine)


def test_fused_adam_matches_torch_adam():
    """ir2rgb_amd.optim.FusedAdam against torch.optim.Adam (the reference's optimizer) over 4 steps on
    tensors of awkward sizes, one of them with a gradient that is a 4-byte-aligned view (scalar path)."""
    from ir2rgb_amd.optim import FusedAdam
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    shapes = [(3,), (64, 3, 7, 7), (8193,), (1024, 130), (1,), (2, 2, 4, 4)]
    init = [torch.randn(s, generator=g) for s in shapes]
    mine = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    ref = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    opt_m = FusedAdam(mine, lr=2e-3, betas=(0.5, 0.999))
    opt_r = torch.optim.Adam(ref, lr=2e-3, betas=(0.5, 0.999))
    flat = torch.empty(sum(t.numel() for t in init) + 1, device=dev)
    for step in range(4):
        grads = [torch.randn(s, generator=g).to(dev) * (10.0 ** (step - 2)) for s in shapes]
        off = 1                                               # misaligned views of a flat buffer
        for p, q, gr in zip(mine, ref, grads):
            view = flat[off:off + gr.numel()].view_as(gr)
            view.copy_(gr)
            p.grad = view if step % 2 else gr.clone()
            q.grad = gr.clone()
            off += gr.numel()
        v0 = mine[1]._version
        opt_m.step()
        opt_r.step()
        assert mine[1]._version > v0, "parameter version must be bumped (packed-weight caches key on it)"
        for i, (p, q) in enumerate(zip(mine, ref)):
            assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), (step, i, (p - q).abs().max().item())
    st = opt_r.state[ref[3]]
    m, v = opt_m.moments(3)
    # torch forms exp_avg with lerp_: same value up to a few fp32 roundings (visible where it cancels)
    dm = ((m - st["exp_avg"]).abs() / (st["exp_avg"].abs() + 1e-3)).max().item()
    dv = ((v - st["exp_avg_sq"]).abs() / (st["exp_avg_sq"].abs() + 1e-6)).max().item()
    assert dm <= 1e-4 and dv <= 1e-4, (dm, dv)   # measured 3e-5 / 1.3e-5: fp32 rounding where the running sums cancel


def test_side_stream_weight_gradients_change_nothing():
    """One training window with the generators' weight gradients on the side HIP stream and one with
    everything on the main stream: discriminator gradients (deterministic kernels only) must be
    bit-identical, generator gradients equal up to the float atomics of the warp backward."""
    from ir2rgb_amd import autograd as AG
    from ir2rgb_amd import vid2vid as V
    dev = _dev()
    A, B = V.synthetic_sequence(4, 64, 128, 7, dev)
    grads = []
    default = AG.WGRAD_SIDE_STREAM
    for side in (True, False):
        AG.WGRAD_SIDE_STREAM = side
        try:
            tr = V.Vid2VidTrainer(dev, seed=0, first_layer_gen_filters=64, gen_blocks=2)
            tr.optimizer_G.step = tr.optimizer_D.step = lambda: None       # keep the gradients, skip the update
            for o in tr.optimizer_D_T:
                o.step = lambda: None
            tr.train_window(A[:, :3], B[:, :3])
            torch.cuda.synchronize()
            grads.append(([p.grad.clone() for p in tr.grads_G.params], [p.grad.clone() for p in tr.grads_D.params]))
        finally:
            AG.WGRAD_SIDE_STREAM = default
    for a, b in zip(grads[0][1], grads[1][1]):
        assert torch.equal(a, b)
    for a, b in zip(grads[0][0], grads[1][0]):
        assert (a - b).norm().item() <= 1e-4 * b.norm().item() + 1e-12


def test_optimizer_step_reaches_the_packed_weights():
    """Two windows with the one-launch Adam and with torch.optim.Adam: same losses in the SECOND window
    (which runs on the weights the first step produced), and every cached packed weight is older than
    its parameter after a step.  Guards the packed-weight cache against an optimizer that updates
    parameters without bumping their version (torch's fused=True Adam does exactly that)."""
    from ir2rgb_amd import vid2vid as V
    dev = _dev()
    A, B = V.synthetic_sequence(5, 64, 128, 7, dev)
    res = []
    for fused in (True, False):
        tr = V.Vid2VidTrainer(dev, seed=0, first_layer_gen_filters=64, gen_blocks=2, fused_adam=fused, batched_repack=False)
        first = {k: v.item() for k, v in tr.train_window(A[:, 0:3], B[:, 0:3]).items()}
        second = {k: v.item() for k, v in tr.train_window(A[:, 1:4], B[:, 1:4]).items()}
        res.append(second)
        assert abs(second["G"] - first["G"]) > 1e-3 * abs(first["G"])
        n_checked = 0
        for net in tr.netG + [tr.netD]:
            for m in net.modules():
                for tag, hit in getattr(m, "_ir2rgb_packed", {}).items():
                    key = hit[0]
                    if tag in ("w", "xexp"):
                        assert m.weight._version > key[2], "optimizer step did not invalidate the packed weight"
                        n_checked += 1
        assert n_checked > 20
    for k in res[0]:
        assert abs(res[0][k] - res[1][k]) <= 5e-3 * abs(res[1][k]), (k, res[0][k], res[1][k])


def test_batched_repack_equals_lazy_repack():
    """layers.WeightRepacker (one launch refreshing every packed weight copy in place after the optimizer
    steps) against the lazy per-layer repack: bit-identical losses over three windows, every refreshed
    buffer bit-identical to a fresh individual pack of the current parameter, and the cache keys current."""
    from ir2rgb_amd import conv as C
    from ir2rgb_amd import vid2vid as V
    dev = _dev()
    A, B = V.synthetic_sequence(6, 64, 128, 7, dev)
    res = []
    for batched in (True, False):
        tr = V.Vid2VidTrainer(dev, seed=0, first_layer_gen_filters=64, gen_blocks=2, batched_repack=batched)
        res.append([{k: v.item() for k, v in tr.train_window(A[:, w:w + 3], B[:, w:w + 3]).items()} for w in range(3)])
        if not batched:
            continue
        n_plain, kinds = 0, set()
        for net in tr.netG + [tr.netD] + tr.netD_T:
            for m in net.modules():
                for tag, hit in getattr(m, "_ir2rgb_packed", {}).items():
                    if len(hit) == 6 and hit[2] is not None:
                        assert hit[0][2] == m.weight._version, (type(m).__name__, tag)
                        fresh = C.pack_weight(hit[2], hit[4], adjoint=hit[3])     # (hit[4]: the parameter, or its rearranged copy)
                        if hit[5] is None:
                            assert hit[4].data_ptr() == m.weight.data_ptr()
                        assert torch.equal(fresh.view(torch.int16), hit[1].view(torch.int16)), (type(m).__name__, tag)
                        n_plain += 1
                        kinds.add((tag, hit[3], hit[2].transposed, hit[2].kh, hit[2].stride_h))
        assert n_plain > 40 and len(kinds) >= 6, (n_plain, kinds)
        assert tr.repacker.batch is not None and sum(b.nentries for b in tr.repacker.batch) >= n_plain
    assert res[0] == res[1]


def test_shared_discriminator_forward_changes_nothing():
    """netD on generated frames evaluated once (serving the D and the G loss through backward flags)
    against the reference's literal three forwards per compute_loss_D: same losses, bit-identical
    discriminator gradients, generator gradients equal up to the float atomics of the warp backward,
    and BatchNorm running statistics advanced as by the literal sequence.  With ``batched_D`` (the default: real |
    generated | raw frames as one batch of sample groups) the convolutions tile and sum a three-sample batch, so the
    same quantities agree to rounding instead of bit for bit."""
    from ir2rgb_amd import vid2vid as V
    dev = _dev()
    A, B = V.synthetic_sequence(9, 64, 128, 7, dev)
    runs = []
    for shared, batched in ((True, False), (False, False), (True, True)):
        tr = V.Vid2VidTrainer(dev, seed=0, first_layer_gen_filters=64, gen_blocks=2, shared_fake_forward=shared, batched_D=batched)
        noop = lambda: None  # noqa: E731  keep the gradients, skip the updates
        tr.optimizer_G.step = tr.optimizer_D.step = noop
        for o in tr.optimizer_D_T:
            o.step = noop
        for w in range(7):      # the 7th window has both temporal scales active
            out = tr.train_window(A[:, w:w + 3], B[:, w:w + 3])
        torch.cuda.synchronize()
        assert "D_T1" in out
        runs.append(({k: v.item() for k, v in out.items()},
                     [p.grad.clone() for p in tr.grads_G.params],
                     [p.grad.clone() for g in [tr.grads_D] + tr.grads_DT for p in g.params],
                     [b.clone() for d in [tr.netD] + tr.netD_T for b in d.buffers()]))
    for k in runs[0][0]:
        assert abs(runs[0][0][k] - runs[1][0][k]) <= 1e-6 * abs(runs[1][0][k]), k
    for a, b in zip(runs[0][2], runs[1][2]):
        assert torch.equal(a, b)
    for a, b in zip(runs[0][1], runs[1][1]):
        assert (a - b).norm().item() <= 1e-4 * b.norm().item() + 1e-12
    for a, b in zip(runs[0][3], runs[1][3]):
        assert torch.equal(a, b), "BatchNorm running statistics / counters differ"
    # batched sample groups against the literal sequence
    dev_l = max(abs(runs[2][0][k] - runs[1][0][k]) / max(abs(runs[1][0][k]), 1e-12) for k in runs[1][0])
    dev_d = max(((a - b).norm() / b.norm().clamp_min(1e-20)).item() for a, b in zip(runs[2][2], runs[1][2]))
    dev_g = max(((a - b).norm() / b.norm().clamp_min(1e-20)).item() for a, b in zip(runs[2][1], runs[1][1]))
    dev_b = max(((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item() for a, b in zip(runs[2][3], runs[1][3]))
    print("batched vs literal: losses %.2e, D gradients %.2e, G gradients %.2e, BatchNorm buffers %.2e" % (dev_l, dev_d, dev_g, dev_b))
    assert dev_l <= 2e-3 and dev_d <= 3e-2 and dev_g <= 3e-2 and dev_b <= 1e-4, (dev_l, dev_d, dev_g, dev_b)


def test_reused_skipped_pair_flows_change_nothing():
    """reuse_skipped_flows: the reference flow / confidence of the older frame pair of a temporally skipped triplet is the
    one FlowNet2 computed for the same two real frames three windows earlier (Vid2VidTrainer.reference_flows) instead of a
    recomputation (discriminator.py:281-283).  Two trainers in lockstep, with and without: the KEPT pair's flow and
    confidence are bit-identical to the recomputed ones in every window.  (The other pairs of a window go through FlowNet2
    in a batch of 2 instead of 3, for which the convolution dispatcher picks other tile forms: same half-precision network,
    other rounding -- up to ~0.2 px on 8 px flows here, with or without reuse -- so they are compared to that level only.)
    Sixteen windows: the reuse starts at window 9 and FlowNet2's HIP graph for the smaller batch replays -- on the second
    stream -- from window 12 on (a kept flow concatenated with a replayed one on the wrong stream went unnoticed in a
    ten-window version of this test).  ``resident_inputs`` (FlowNet2 ahead of the main stream) only reorders streams: every
    loss stays bit-identical."""
    from ir2rgb_amd import vid2vid as V
    dev = _dev()
    A, B = V.synthetic_sequence(18, 64, 128, 3, dev)
    trs, caps, losses = [], [], []
    for reuse, resident in ((True, False), (False, False), (True, True)):
        tr = V.Vid2VidTrainer(dev, seed=0, first_layer_gen_filters=64, gen_blocks=2, reuse_skipped_flows=reuse, resident_inputs=resident)
        cap = []
        orig = tr.reference_flows

        def wrapped(*a, _orig=orig, _cap=cap, **k):
            out = _orig(*a, **k)
            _cap.append(out)
            return out
        tr.reference_flows = wrapped
        trs.append(tr)
        caps.append(cap)
        losses.append([])
    n_kept = 0
    for w in range(16):
        for tr, ls in zip(trs, losses):
            ls.append({k: v.item() for k, v in tr.train_window(A[:, w:w + 3], B[:, w:w + 3]).items()})
        torch.cuda.synchronize()
        (f0, c0, _, e0), (f1, c1, _, e1) = caps[0][-1], caps[1][-1]
        assert (f0 - f1).abs().max().item() <= 0.5 and (f0 - f1).abs().mean().item() <= 0.1
        if 1 in e1:
            assert e0[1][0].shape == e1[1][0].shape == (1, 2, 2, 64, 128)
            if w >= 9:      # pair 0 of the reusing trainer is the kept one
                assert torch.equal(e0[1][0][:, 0], e1[1][0][:, 0]) and torch.equal(e0[1][1][:, 0], e1[1][1][:, 0]), w
                n_kept += 1
            assert (e0[1][0][:, 1] - e1[1][0][:, 1]).abs().max().item() <= 0.5
    assert n_kept == 7 and "D_T1" in losses[0][-1]
    assert trs[2]._early_on                     # the last windows really ran FlowNet2 ahead of the main stream
    assert losses[2] == losses[0]               # ... and that changed nothing
