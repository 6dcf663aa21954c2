"""CPU tests of the device-side frame bookkeeping (ir2rgb_amd/frames.py): FrameHistory.push
against a literal evaluation of the reference's get_skipped_frames (models/discriminator.py:257-271) over long push
sequences (buffer wrap-arounds included), with autograd through the tuples that contain current frames; WindowSlicer
windows as zero-copy views."""
import pytest
import torch

from ir2rgb_amd.frames import FrameHistory, WindowSlicer


def get_skipped_frames(B_all, B, t_scales, tD):
    """discriminator.py:257-271, evaluated as written (the reference module itself cannot be imported: cv2)."""
    B_all = torch.cat([B_all.detach(), B], dim=1) if B_all is not None else B
    B_skipped = [None] * t_scales
    for s in range(t_scales):
        tDs = tD ** s
        span = tDs * (tD - 1)
        n_groups = min(B_all.size()[1] - span, B.size()[1])
        if n_groups > 0:
            for t in range(0, n_groups, tD):
                skip = B_all[:, (-span - t - 1):-t:tDs].contiguous() if t != 0 else B_all[:, -span - 1::tDs].contiguous()
                B_skipped[s] = torch.cat([B_skipped[s], skip]) if B_skipped[s] is not None else skip
    max_prev_frames = tD ** (t_scales - 1) * (tD - 1)
    if B_all.size()[1] > max_prev_frames:
        B_all = B_all[:, -max_prev_frames:]
    return B_all, B_skipped


@pytest.mark.parametrize("t_scales,tD,n_new,batch", [(2, 3, 1, 1), (3, 3, 1, 2), (2, 3, 2, 1), (2, 3, 4, 1), (1, 3, 1, 1), (3, 2, 3, 2)])
def test_frame_history_matches_get_skipped_frames(t_scales, tD, n_new, batch):
    g = torch.Generator().manual_seed(t_scales * 100 + tD * 10 + n_new)
    hist, B_all = FrameHistory(t_scales, tD), None
    for step in range(40):                     # far past the buffer capacity: several wrap-arounds
        B = torch.randn(batch, n_new, 2, 3, 4, generator=g)
        B_all, want = get_skipped_frames(B_all, B, t_scales, tD)
        got = hist.push(B)
        for s in range(t_scales):
            assert (got[s] is None) == (want[s] is None), (step, s)
            if want[s] is not None:
                assert got[s].shape == want[s].shape and got[s].is_contiguous() and torch.equal(got[s], want[s]), (step, s)
        assert torch.equal(hist.frames(), B_all)
    assert hist.buf.shape[1] == 2 * (hist.keep + n_new)          # never re-allocated


def test_frame_history_keeps_the_gradient_of_current_frames():
    hist, B_all = FrameHistory(2, 3), None
    for step in range(9):
        B = torch.randn(1, 1, 2, 3, 3, requires_grad=True)
        Br = B.detach().clone().requires_grad_()
        B_all, want = get_skipped_frames(B_all, Br, 2, 3)
        got = hist.push(B)
        loss_w = sum((w * (i + 1)).sum() for i, w in enumerate(want) if w is not None)
        loss_g = sum((w * (i + 1)).sum() for i, w in enumerate(got) if w is not None)
        if isinstance(loss_w, torch.Tensor):
            loss_w.backward()
            loss_g.backward()
            assert torch.equal(B.grad, Br.grad)
    assert not hist.frames().requires_grad


def test_frame_history_load_and_reset():
    hist = FrameHistory(2, 3)
    frames = torch.arange(10 * 2.0).view(1, 10, 2, 1, 1)
    hist.load(frames)
    assert torch.equal(hist.frames(), frames[:, -6:])
    out = hist.push(torch.full((1, 1, 2, 1, 1), -1.0))
    assert out[1] is not None and torch.equal(out[1][0, :, 0, 0, 0], torch.tensor([8.0, 14.0, -1.0]))   # frames 4, 7, new
    hist.reset()
    assert hist.push(torch.zeros(1, 1, 2, 1, 1)) == [None, None]


@pytest.mark.parametrize("n_frames,tg,n_load", [(30, 3, 1), (30, 3, 2), (6, 3, 4), (12, 2, 3), (2, 3, 1)])
def test_window_slicer(n_frames, tg, n_load):
    cin, cout, h, w = 3, 2, 4, 5
    ir = torch.arange(n_frames * cin * h * w, dtype=torch.float32).view(1, n_frames * cin, h, w)
    rgb = -torch.arange(n_frames * cout * h * w, dtype=torch.float32).view(1, n_frames * cout, h, w)
    ws = WindowSlicer(ir, rgb, n_input_gen_frames=tg, n_frames_load=n_load, input_nc=cin, output_nc=cout)
    assert len(ws) == max((n_frames - tg + 1) // n_load, 0)
    for i, (a, b) in enumerate(ws):
        assert a.shape == (1, n_load + tg - 1, cin, h, w) and b.shape == (1, n_load + tg - 1, cout, h, w)
        assert a.data_ptr() == ir.data_ptr() + 4 * i * n_load * cin * h * w          # a view, not a copy
        assert torch.equal(a[0, 0], ir[0, i * n_load * cin:(i * n_load + 1) * cin])
        assert torch.equal(b[0, -1], rgb[0, (i * n_load + n_load + tg - 2) * cout:(i * n_load + n_load + tg - 1) * cout])
        # consecutive windows overlap by the tG - 1 frames the generator conditions on
        if i:
            assert torch.equal(a[:, :tg - 1], prev_a[:, -(tg - 1):])
        prev_a = a
    with pytest.raises(IndexError):
        ws[len(ws)]
    with pytest.raises(ValueError):
        WindowSlicer(ir[:, :-1], rgb)
