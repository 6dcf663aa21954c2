"""Concurrency tests: kernels of this library next to each other on two HIP streams.

Round 2 left a corruption on record that only showed when a FRESH generator ran its first forward on two streams.  Cause
(round 3): a write-after-read race inside the convolution kernels' LDS rings -- hipcc had sunk the wait for
a K-step's last fragment reads below the raw s_barrier behind which the other waves restage that buffer by LDS-DMA; the
read loses that race only when something slows the LDS, which the other stream's bank-conflicted weight-packing kernel on
the same CU did.  These tests hold the trigger in isolation and the original scenario; tests/test_abi.py holds the static
check of the code (tools/check_lds_war.py)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

OPT = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)

CASES = {
    # name: (dtype, Cin, H, W, Cout, k, stride, pad, pad_mode name)
    "igemm_2stage_128ch_3x3": (torch.float16, 128, 512, 1024, 128, 3, 1, 1, "PAD_REFLECT"),
    "igemm_3stage_512ch_3x3s2": (torch.bfloat16, 512, 128, 256, 1024, 3, 2, 1, "PAD_ZERO"),
    "patch_large_1024ch_64x128": (torch.bfloat16, 1024, 64, 128, 1024, 3, 1, 1, "PAD_REFLECT"),
    "patch_adj_split_1024ch_32x64": (torch.bfloat16, 1024, 32, 64, 1024, 3, 1, 1, "PAD_REFLECT_ADJ"),
}


@pytest.mark.parametrize("name", list(CASES))
def test_convolutions_are_exact_beside_an_lds_heavy_kernel(dev, name):
    """Forty launches of a convolution while the tiled weight-packing kernel (19 KB of LDS per workgroup, bank-conflicted
    2-byte reads) runs on a second stream: every launch must equal the solo launch bit for bit.  The 2-stage conv_igemm
    case failed 10-40 % of its launches before the fix (tools/lds_war_stress.py with the library from before it)."""
    from ir2rgb_amd import conv as C
    dtype, cin, h, w, cout, k, stride, pad, pm = CASES[name]
    pad_mode = getattr(C, pm)
    gen = torch.Generator().manual_seed(len(name))
    x = torch.randn(1, cin, h, w, generator=gen).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, k, k, generator=gen) * 0.05).to(dev)
    desc = C.make_desc(tuple(x.shape), cout, k, stride, pad, pad_mode, dtype)
    if pad_mode == C.PAD_REFLECT_ADJ:
        wp = C.pack_weight(C.make_desc(tuple(x.shape), cout, k, stride, pad, C.PAD_ZERO, dtype), wt, adjoint=True)
    else:
        wp = C.pack_weight(desc, wt)
    ref, _ = C.conv2d_fwd(desc, x, wp)
    hw = (torch.randn(1024, 1024, 3, 3, generator=gen) * 0.05).to(dev)
    hdesc = C.make_desc((1, 1024, 32, 64), 1024, 3, 1, 1, C.PAD_ZERO, torch.bfloat16)
    side = torch.cuda.Stream(dev)
    outs = [torch.empty_like(ref) for _ in range(4)]
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for i in range(40):
        with torch.cuda.stream(side):
            for _ in range(3):
                C.pack_weight(hdesc, hw)
        y, _ = C.conv2d_fwd(desc, x, wp, out=outs[i % 4])
        bad += (y != ref).any()
    torch.cuda.synchronize()
    assert int(bad) == 0, f"{int(bad)} of 40 launches differ from the solo launch"


@pytest.mark.parametrize("model", ["composite", "composite-local"])
def test_first_forward_of_a_fresh_module_on_two_streams(dev, model):
    """The original scenario: a FRESH generator (weights still to be packed, caches to be built) runs its first forward
    with its branches on two streams, after earlier forwards have filled the allocator's caches; its deep copy runs on
    one stream.  Outputs bit-identical, four fresh modules in a row."""
    from ir2rgb_amd import networks as N
    local = model == "composite-local"
    H, W = (1024, 2048) if local else (512, 1024)
    gen = torch.Generator().manual_seed(1)
    A, P = torch.rand(1, 9, H, W, generator=gen).to(dev), torch.rand(1, 6, H, W, generator=gen).to(dev)
    fi = ff = None
    if local:
        mk = lambda: torch.rand(1, 128, H // 2, W // 2, generator=gen).to(dev).half().contiguous(memory_format=torch.channels_last)  # noqa: E731
        fi, ff = mk(), mk()
    for it in range(4):
        torch.manual_seed(0)
        ga = N.build_generator_module(9, 3, 6, 64 if local else 128, model, 3, "batch", 1 if local else 0, **OPT).to(dev).train()
        gb = copy.deepcopy(ga)
        ga.compute_dtype = gb.compute_dtype = torch.float16
        torch.cuda.synchronize()
        with torch.no_grad():
            with N.branch_streams():
                two = ga(A, P, None, fi, ff, None, False)[:6]
            torch.cuda.synchronize()
            with N.branch_streams(False):
                one = gb(A, P, None, fi, ff, None, False)[:6]
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(one, two)), f"fresh module {it}"
        del ga, gb, one, two


def test_stream_check_catches_an_unordered_use_and_accepts_an_ordered_one(dev):
    """ir2rgb_amd.streamcheck (IR2RGB_STREAM_CHECK=1): a cached buffer written on one stream and read on another without a
    wait in between raises; with wait_stream / an event / a host synchronisation in between it does not."""
    from ir2rgb_amd import streamcheck as SC
    SC.enable()
    try:
        side = torch.cuda.Stream(dev)
        buf = torch.zeros(16, device=dev)
        with torch.cuda.stream(side):
            buf.add_(1)
            SC.produced(buf, "planted buffer")
        with pytest.raises(SC.StreamOrderError):
            SC.consumed(buf)
        torch.cuda.current_stream(dev).wait_stream(side)
        SC.consumed(buf)                                   # ordered by wait_stream
        with torch.cuda.stream(side):
            buf.add_(1)
            SC.produced(buf, "planted buffer")
            ev = torch.cuda.Event()
            ev.record()
        with pytest.raises(SC.StreamOrderError):
            SC.consumed(buf)
        torch.cuda.current_stream(dev).wait_event(ev)
        SC.consumed(buf)                                   # ordered by the event
        with torch.cuda.stream(side):
            buf.add_(1)
            SC.produced(buf, "planted buffer")
        torch.cuda.synchronize()
        SC.consumed(buf)                                   # ordered by the host
        third = torch.cuda.Stream(dev)
        with torch.cuda.stream(third):
            SC.consumed(buf)                               # a stream created after the host wait is ordered after it too
    finally:
        SC.disable()
    assert torch.cuda.Stream.wait_stream.__name__ != "wait_stream" or not SC.ENABLED


def test_sixteen_training_windows_under_the_stream_check(dev):
    """The whole loop on its three streams (main; FlowNet2 replaying ahead of it beside the previous window's optimizer
    step; kept pair flows crossing windows and streams; the batched repack) with every cached buffer tagged by its producing
    stream: no un-ordered cross-stream use in sixteen windows, and the check really saw cross-stream uses."""
    from ir2rgb_amd import streamcheck as SC
    from ir2rgb_amd import vid2vid as V
    SC.enable()
    try:
        A, B = V.synthetic_sequence(18, 64, 128, 3, dev)
        tr = V.Vid2VidTrainer(dev, seed=0, first_layer_gen_filters=64, gen_blocks=2, resident_inputs=True)
        for w in range(16):
            out = tr.train_window(A[:, w:w + 3], B[:, w:w + 3])
        torch.cuda.synchronize()
        assert tr._early_on and all(torch.isfinite(v).all() for v in out.values())
        assert SC.STATS["cross_stream"] > 0 and SC.STATS["produced"] > 1000, SC.STATS
        # ... and an inference forward on two streams right after the training step that rewrote the weights
        with torch.no_grad():
            b, t, c, h, w_ = A[:, :3].shape
            tr.netG[0](A[:, :3].reshape(b, -1, h, w_), B[:, :2].reshape(b, -1, h, w_), None, None, None, None, False)
        torch.cuda.synchronize()
    finally:
        SC.disable()


def test_discriminator_streams_change_nothing_but_the_schedule(dev):
    """The image discriminator's scales and the temporal discriminators on their own HIP streams (forward; the backward
    passes follow through autograd) against the one-stream trainer: same kernels on the same operands, so the first window's
    loss terms agree to the order in which the engine adds the three discriminators' gradient contributions of a generated
    frame (the streams change the order the graph is built in), later windows to what training makes of such a difference
    -- and two runs WITH the streams agree bit for bit, losses and parameters."""
    from ir2rgb_amd import vid2vid as V
    A, B = V.synthetic_sequence(14, 64, 128, 5, dev)

    def run(streams):
        tr = V.Vid2VidTrainer(dev, seed=0, first_layer_gen_filters=64, gen_blocks=2, resident_inputs=True,
                              discriminator_streams=streams)
        assert tr.d_streams == streams and getattr(tr.netD, "scale_streams", False) == streams
        outs = [tr.train_window(A[:, w:w + 3], B[:, w:w + 3]) for w in range(12)]
        torch.cuda.synchronize()
        params = torch.cat([p.detach().reshape(-1) for m in tr.netG + [tr.netD] + tr.netD_T for p in m.parameters()])
        return [{k: float(v) for k, v in o.items()} for o in outs], params

    on1, p1 = run(True)
    on2, p2 = run(True)
    assert on1 == on2 and torch.equal(p1, p2), "two runs with discriminator streams differ"
    off, p0 = run(False)
    assert set(on1[-1]) == set(off[-1]) and any(k.startswith("D_T") for k in off[-1])
    # the first window differs by summation order only; later ones by what twelve adversarial updates make of that
    # (measured: 4e-3 on the D loss by window 8), so the bound loosens with the window and ends as a sanity check
    dev_by_window = [max(abs(a[k] - b[k]) / (abs(b[k]) + 1e-3) for k in b) for a, b in zip(on1, off)]
    assert dev_by_window[0] <= 1e-4, dev_by_window
    assert max(dev_by_window[:3]) <= 5e-3, dev_by_window
    assert max(dev_by_window) <= 0.2, dev_by_window
    assert all(v == v and abs(v) < 1e6 for o in on1 for v in o.values())
